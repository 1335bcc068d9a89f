#!/bin/bash
R=$PWD; O=$R/gpurun_out/r02q; mkdir -p $O
timeout -k 10 400 python -u -m pytest tests -m gpu -x -q --timeout 200 --timeout-method thread -k "emission or emit or photons or lists or sharded or preset" > $O/tests.txt 2>&1 || { tail -30 $O/tests.txt; exit 1; }
tail -2 $O/tests.txt
timeout -k 10 300 python bench.py --no-cpu-baseline --no-gather --steps 50 > $O/bench.json 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
python - <<PY
import json
d=json.load(open("$O/bench.json"))
print(json.dumps(d["roofline_emit"], indent=1)[:900]); print(json.dumps(d["extra"]["photon_emission"], indent=1)); print(d["extra"].get("full_mode_b_frame"))
PY
