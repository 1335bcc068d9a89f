#!/bin/bash
R=$PWD; O=$R/gpurun_out/r02m; mkdir -p $O
timeout -k 10 300 python -u -m pytest tests -m gpu -x -q --timeout 200 --timeout-method thread -k "pipelines_agree or golden or frame" > $O/tests.txt 2>&1 || { tail -30 $O/tests.txt; exit 1; }
tail -2 $O/tests.txt
for cfg in "0 1" "16 1" "32 1" "64 1" "32 2" "64 2" "64 3"; do set -- $cfg; for n in 3; do
  RT355_TAIL_HANDOFF=$1 RT355_TAIL_ROUNDS=$2 RT355_BENCH_IN_FLIGHT=$n timeout -k 10 120 python bench.py --no-cpu-baseline --no-gather --no-other-configs --steps 400 --warmup 10 > $O/h.json 2> $O/h.err || { tail -5 $O/h.err; exit 1; }
  python - <<PY
import json
d=json.load(open("$O/h.json"))
print("handoff $1 rounds $2,", $n, "in flight:", round(d["value"]), "Mrays/s", round(d["ms_per_step"],4), "ms/frame; frame frac", round(d["roofline"]["frame"]["frac"],4), "tail excl", round(d["roofline"]["exclusive"]["tail"]["avg_launch_ms"],4))
PY
done; done
