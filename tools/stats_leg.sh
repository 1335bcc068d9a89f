#!/bin/bash
# kernel-time table of one bench leg (rocprofv3 --kernel-trace --stats), printed; run on the GPU box: tools/stats_leg.sh full_b [pattern]
R=$PWD; cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/stats_leg; rm -rf $O; mkdir -p $O
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O -- python3 $R/bench.py --no-cpu-baseline --steps 8 --warmup 2 --legs $1 > $O/bench.json 2> $O/err.txt || exit 1
cd $R
python3 tools/prof_summary.py $(ls -t $(find $O -name "*.db") | head -1) "$1" | cut -c1-80,100-175 | grep -E "${2:-.}" | head -${3:-30}
rm -rf $O
