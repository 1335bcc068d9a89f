# round-end rehearsal: the default bench command, timed
set -e
S=$(date +%s)
timeout -k 10 600 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
echo "bench wall $(( $(date +%s) - S )) s" >> gpurun_out/bench_default.err
