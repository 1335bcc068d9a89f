# GPU tests, then the frame leg of the bench with serial and with overlapping frames (experiment)
set -e
timeout -k 10 500 python -u -m pytest tests -m gpu -x -q --timeout 180 --timeout-method thread > gpurun_out/gpu_tests.log 2>&1
RT355_BENCH_SERIAL_FRAMES=1 timeout -k 10 300 python bench.py --no-cpu-baseline --no-other-configs --no-gather > gpurun_out/exp_serial.json 2> gpurun_out/exp_serial.err
timeout -k 10 300 python bench.py --no-cpu-baseline --no-other-configs --no-gather > gpurun_out/exp_overlap.json 2> gpurun_out/exp_overlap.err
RT355_BENCH_NO_EVENTS=1 timeout -k 10 300 python bench.py --no-cpu-baseline --no-gather --no-other-configs > gpurun_out/exp_overlap_noev.json 2> gpurun_out/exp_overlap_noev.err
