#!/bin/bash
# Profiles of the bench command for profiles/: kernel-time statistics, then HBM traffic and SQ counters in
# passes of their own (counters are never combined with other trace domains).  Run on the GPU box from the repo root.
set -e
R=$PWD
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/prof
rm -rf $O && mkdir -p $O
B="python3 $R/bench.py --no-cpu-baseline --steps 5 --warmup 2"
( while sleep 45; do echo "[profile_round] $(date +%T) running"; done ) &
HB=$!
trap "kill $HB 2>/dev/null" EXIT
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/stats -- $B > $O/stats.json 2> $O/stats.err
timeout -s KILL 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/fetch -- $B > $O/fetch.json 2> $O/fetch.err
timeout -s KILL 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/write -- $B > $O/write.json 2> $O/write.err
timeout -s KILL 400 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY -d $O/sq -- $B > $O/sq.json 2> $O/sq.err
ls $O
