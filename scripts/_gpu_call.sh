set -o pipefail
mkdir -p gpurun_out/r05j
bash scripts/gpu_profile.sh r05 > gpurun_out/r05j/profile.log 2>&1; echo "profile rc=$?"
bash scripts/gpu_profile.sh r05s --variant smooth > gpurun_out/r05j/profile_smooth.log 2>&1; echo "profile smooth rc=$?"
timeout -k 10 700 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05j/bench.json 2> gpurun_out/r05j/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
b=json.loads(open('gpurun_out/r05j/bench.json').read().strip().splitlines()[-1])
print({k:v for k,v in b.items() if not isinstance(v,(dict,list))})
print(json.dumps(b['config'].get('depth_calibration')))
print(json.dumps({k:{kk:vv for kk,vv in v.items() if kk in ('streams','stage_pipeline','tail_from','ms_per_step','speedup_vs_1','lone_job_ms','efficiency','lone_job_issue_floor_frac')} for k,v in b['strong_projection']['by_n_gpus'].items()}))
PY
timeout -k 10 900 python -m pytest tests/test_distributed_gloo.py -m gpu -x -q > gpurun_out/r05j/gputests.log 2>&1; echo "pytest rc=$?"
tail -4 gpurun_out/r05j/gputests.log
