set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/pmc6
timeout -s KILL 60 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_WAIT_ANY -d $R/gpurun_out/pmc6 -- python3 $R/bench.py --no-cpu-baseline --no-gather --no-other-configs --steps 3 --warmup 1 > $R/gpurun_out/pmc6.json 2> $R/gpurun_out/pmc6.err
