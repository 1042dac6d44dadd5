set -o pipefail
mkdir -p gpurun_out/r05k
timeout -k 10 400 python tests/tools/soak_parity.py --cases 40000 --seed 105 > gpurun_out/r05k/soak_parity.json 2> gpurun_out/r05k/soak_parity.err; echo "soak parity rc=$?"; tail -c 600 gpurun_out/r05k/soak_parity.json
timeout -k 10 400 python tests/tools/soak_chunks.py --cases 3000 --seed 106 > gpurun_out/r05k/soak_chunks.json 2> gpurun_out/r05k/soak_chunks.err; echo "soak chunks rc=$?"; tail -c 600 gpurun_out/r05k/soak_chunks.json
timeout -k 10 400 python tests/tools/soak_generic.py --legs 256 --seed 107 --queue > gpurun_out/r05k/soak_generic.json 2> gpurun_out/r05k/soak_generic.err; echo "soak generic rc=$?"; tail -c 800 gpurun_out/r05k/soak_generic.json
