set -e
timeout -k 10 400 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/gpu_tests.log 2>&1
RT355_KNN_VERBOSE=1 RT355_LIB=$PWD/gpurun_variants/librt355_knnprobe.so timeout -k 10 200 python bench.py --no-cpu-baseline --steps 3 > gpurun_out/exp_probe.json 2> gpurun_out/exp_probe.err
timeout -k 10 200 python bench.py --no-cpu-baseline > gpurun_out/exp_full.json 2> gpurun_out/exp_full.err
