#!/bin/bash
# 2-rank dry run of `bench.py --gpus 2` on a one-GPU box: both ranks share GPU 0 and talk over gloo (the numbers mean
# nothing; the control flow, the shard exchange and the three split phases are the real ones)
set -e
mkdir -p gpurun_out
RT355_BENCH_REHEARSAL=1 timeout -k 10 ${LIMIT:-500} python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 \
  bench.py --gpus 2 --steps ${STEPS:-8} --warmup 2 --photons-per-gpu ${PHOTONS:-1000000} --no-gather --no-other-configs > gpurun_out/rehearsal_2ranks.json 2> gpurun_out/rehearsal_2ranks.err
tail -c 3000 gpurun_out/rehearsal_2ranks.json
