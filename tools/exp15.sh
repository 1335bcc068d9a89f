# default bench command three times: run-to-run spread (experiment)
set -e
for i in 1 2 3; do
  timeout -k 10 600 python bench.py --no-cpu-baseline --no-gather --no-other-configs > gpurun_out/exp_rep$i.json
done
