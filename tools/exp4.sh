# bench (frame leg) with 0, 1, 2 single-bounce passes, then a kernel trace of the 2-pass schedule (experiment)
set -e
R=$GRAFT_REPO_ROOT
for n in 0 1 2; do
  RT355_BOUNCE_PASSES=$n RT355_BENCH_NO_EVENTS=1 timeout -k 10 120 python bench.py --no-cpu-baseline --no-gather --no-other-configs > gpurun_out/exp_b$n.json
done
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/kt1
RT355_BENCH_NO_EVENTS=1 timeout -k 10 60 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/kt1 -- python3 $R/bench.py --no-cpu-baseline --no-gather --no-other-configs --steps 5 --warmup 1 > $R/gpurun_out/kt1.json 2> $R/gpurun_out/kt1.err
