set -o pipefail
mkdir -p gpurun_out/r05g
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r05g/gputests.log 2>&1; echo "pytest rc=$?"
tail -4 gpurun_out/r05g/gputests.log
timeout -k 10 700 python bench.py --steps 20 --warmup 5 > gpurun_out/r05g/bench.json 2> gpurun_out/r05g/bench.err; echo "bench rc=$?"
tail -3 gpurun_out/r05g/bench.err
python - <<'PY'
import json
b=json.loads(open('gpurun_out/r05g/bench.json').read().strip().splitlines()[-1])
print({k:v for k,v in b.items() if not isinstance(v,(dict,list))})
print(json.dumps(b['config'].get('depth_calibration')))
print(json.dumps({k:{kk:vv for kk,vv in v.items() if kk in ('streams','stage_pipeline','ms_per_step','speedup_vs_1','lone_job_ms','efficiency')} for k,v in b['strong_projection']['by_n_gpus'].items()}))
print(json.dumps(b['configs']['4']['head_kernel']))
PY
bash scripts/gpu_profile.sh r05 > gpurun_out/r05g/profile.log 2>&1; echo "profile rc=$?"
bash scripts/gpu_profile.sh r05s --variant smooth > gpurun_out/r05g/profile_smooth.log 2>&1; echo "profile smooth rc=$?"
bash scripts/gpu_latency_profile.sh r05 > gpurun_out/r05g/latprofile.log 2>&1; echo "latency profile rc=$?"
bash scripts/gpu_head_profile.sh r05 16000000 > gpurun_out/r05g/headprofile.log 2>&1; echo "head profile rc=$?"
