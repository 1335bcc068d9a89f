set -e
timeout -k 10 400 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/gpu_tests.log 2>&1
RT355_KNN_VERBOSE=1 timeout -k 10 200 python bench.py --no-cpu-baseline > gpurun_out/exp_full.json 2> gpurun_out/exp_full.err
