set -e
timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/exp_full.json 2> gpurun_out/exp_full.err
RT355_BENCH_NO_EVENTS=1 timeout -k 10 300 python bench.py --no-cpu-baseline --no-gather > gpurun_out/exp_noev.json 2> gpurun_out/exp_noev.err
