#!/bin/bash
# SQ counter pass (one rocprofv3 --pmc run) over a python script:  tools/pmc_run.sh <out-name> <script.py> [args]
# with RT355_LIB exported beforehand to pick a library variant.  Prints the per-kernel table (largest dispatch of each).
set -e
R=$PWD
name=$1; shift
O=$R/gpurun_out/pmc_$name
rm -rf $O && mkdir -p $O
cd /tmp && export TMPDIR=/tmp
COUNTERS=${COUNTERS:-"SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY"}
timeout -s KILL ${LIMIT:-240} rocprofv3 --kernel-trace --pmc $COUNTERS -d $O -- python3 $R/"$@" > $O/stdout.txt 2> $O/stderr.txt
db=$(find $O -name "*.db" | head -1)
python3 $R/tools/pmc_summary.py $db "$name" > $R/gpurun_out/pmc_$name.md
grep -E "${FILTER:-k_}" $R/gpurun_out/pmc_$name.md | cut -c1-330 | head -${ROWS:-12}
