# two-stage tail: hand-over threshold x frames in flight (experiment)
set -e
for h in 0 16 32 48 64; do
  for n in 2 3; do
    RT355_TAIL_HAND_OVER=$h RT355_BENCH_IN_FLIGHT=$n RT355_BENCH_NO_EVENTS=1 timeout -k 10 120 python bench.py --no-cpu-baseline --no-gather --no-other-configs --steps 40 > gpurun_out/exp_h${h}_n$n.json
  done
done
