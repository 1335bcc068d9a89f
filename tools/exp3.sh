set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/pmc4 $R/gpurun_out/pmc5
timeout -s KILL 60 rocprofv3 --kernel-trace --pmc SQ_IFETCH SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU -d $R/gpurun_out/pmc4 -- python3 $R/bench.py --no-cpu-baseline --no-gather --steps 3 --warmup 1 > $R/gpurun_out/pmc4.json 2> $R/gpurun_out/pmc4.err
timeout -s KILL 60 rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQC_TC_INST_REQ SQC_TC_STALL -d $R/gpurun_out/pmc5 -- python3 $R/bench.py --no-cpu-baseline --no-gather --steps 3 --warmup 1 > $R/gpurun_out/pmc5.json 2> $R/gpurun_out/pmc5.err
