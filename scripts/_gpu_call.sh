set -o pipefail
mkdir -p gpurun_out/r05n
timeout -k 10 1000 python -m pytest tests/test_distributed_gloo.py -m gpu -x -q > gpurun_out/r05n/gputests.log 2>&1; echo "pytest rc=$?"
tail -15 gpurun_out/r05n/gputests.log
