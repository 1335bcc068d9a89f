# two counter passes over the frame leg of the bench (experiment): instruction cache, LDS, scalar unit
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/pmc7 $R/gpurun_out/pmc8
timeout -s KILL 60 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_IFETCH SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_SCA SQ_INSTS_SALU -d $R/gpurun_out/pmc7 -- python3 $R/bench.py --no-cpu-baseline --no-gather --no-other-configs --steps 3 --warmup 1 > $R/gpurun_out/pmc7.json 2> $R/gpurun_out/pmc7.err
timeout -s KILL 60 rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_ACTIVE_INST_ANY SQ_INSTS_BRANCH SQ_INSTS_VALU SQ_ACTIVE_INST_VALU -d $R/gpurun_out/pmc8 -- python3 $R/bench.py --no-cpu-baseline --no-gather --no-other-configs --steps 3 --warmup 1 > $R/gpurun_out/pmc8.json 2> $R/gpurun_out/pmc8.err
