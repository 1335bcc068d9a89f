#!/bin/bash
# VALU issue-rate micro-benchmark, then kernel statistics of the full bench (incl. the emission leg)
set -e
R=$PWD
mkdir -p $R/gpurun_out/r02a
timeout -k 10 120 $R/tools/ubench/valu_peak > $R/gpurun_out/r02a/valu_peak.txt 2>&1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r02a/stats -- python3 $R/bench.py --no-cpu-baseline --steps 5 --warmup 2 > $R/gpurun_out/r02a/bench.json 2> $R/gpurun_out/r02a/bench.err
ls $R/gpurun_out/r02a
