#!/bin/bash
for i in 1 2; do python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extras --detail-path gpurun_out/r06_cal_$i.json 2>/dev/null | cut -c1-330; done
python - <<'PY'
import json
for i in (1,2):
    d=json.load(open("gpurun_out/r06_cal_%d.json"%i))
    print([(c["streams"],c["chain_queue"],round(c["ms_per_step"],3)) for c in d["config"]["depth_calibration"]["candidates"]], d["config"]["depth_calibration"]["chosen"])
PY
