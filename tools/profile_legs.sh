#!/bin/bash
# Per-leg profiles of the bench command: for every leg named on the command line (frames = the headline frames alone,
# otherwise a --legs name of bench.py) one kernel-time pass and four counter passes, each a rocprofv3 run of its own
# (counters are never combined with other trace domains, at most 8 SQ counters a pass).
#   usage (on the GPU box, from the repo root): tools/profile_legs.sh <tag> frames gather emit ...
# The databases (gpurun_out/prof/<leg>/<pass>/, hundreds of MB) stay on the box: tools/make_leg_profiles.py turns them into
# the tables of gpurun_out/profiles_out/<tag>_*, which are then copied into profiles/.
set -e
R=$PWD
cd /tmp && export TMPDIR=/tmp
( while sleep 45; do echo "[profile_legs] $(date +%T) running"; done ) &
HB=$!
trap "kill $HB 2>/dev/null" EXIT
TAG=$1; shift
for LEG in "$@"; do
  O=$R/gpurun_out/prof/$LEG
  rm -rf $O && mkdir -p $O
  L=$LEG; [ $LEG = frames ] && L=none
  B="python3 $R/bench.py --no-cpu-baseline --steps 8 --warmup 2 --legs $L"
  echo "[profile_legs] $LEG: $B"
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/stats -- $B > $O/stats.json 2> $O/stats.err
  timeout -s KILL 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/fetch -- $B > $O/fetch.json 2> $O/fetch.err
  timeout -s KILL 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/write -- $B > $O/write.json 2> $O/write.err
  timeout -s KILL 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY -d $O/sq -- $B > $O/sq.json 2> $O/sq.err
  timeout -s KILL 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 -d $O/flops -- $B > $O/flops.json 2> $O/flops.err
done
cd $R
PROFILES_OUT=$R/gpurun_out/profiles_out python3 tools/make_leg_profiles.py $TAG "$@"
rm -rf $R/gpurun_out/prof
