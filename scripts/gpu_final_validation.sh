#!/bin/bash
# final validation of the round on the GPU box: the whole GPU tier, smoke(), the soaks on this build, the driver's bench command
set -o pipefail
python -m pytest tests -m gpu -q --durations=8 > gpurun_out/gpu_tests.log 2>&1; echo "pytest rc $?"; tail -3 gpurun_out/gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python tests/tools/soak_parity.py --cases 60000 --seed 11 > gpurun_out/r06_soak_parity.json 2> gpurun_out/r06_soak_parity.err; cut -c1-400 gpurun_out/r06_soak_parity.json
python tests/tools/soak_chunks.py --cases 6000 --seed 5 > gpurun_out/r06_soak_chunks.json 2> gpurun_out/r06_soak_chunks.err; cut -c1-400 gpurun_out/r06_soak_chunks.json
python tests/tools/soak_generic.py > gpurun_out/r06_soak_generic.json 2> gpurun_out/r06_soak_generic.err; cut -c1-400 gpurun_out/r06_soak_generic.json
python tests/tools/soak_queue.py > gpurun_out/r06_soak_queue.json 2> gpurun_out/r06_soak_queue.err; cut -c1-200 gpurun_out/r06_soak_queue.json
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06_bench_final.json 2> gpurun_out/r06_bench_final.err; cp bench_detail.json gpurun_out/r06_bench_final_detail.json; cat gpurun_out/r06_bench_final.json
