set -e
mkdir -p gpurun_out/s12
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/s12/gpu_tests.log 2>&1
timeout -k 10 900 python -u bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/s12/bench.log 2> gpurun_out/s12/bench.err
