# hierarchy tests, then the 10 000-primitive 4K frame (experiment)
set -e
timeout -k 10 500 python -u -m pytest tests -m gpu -x -q --timeout 180 --timeout-method thread -k "hierarchy" > gpurun_out/gpu_tests_h.log 2>&1
timeout -k 10 300 python tools/exp_c4.py > gpurun_out/c4.json 2> gpurun_out/c4.err
