#!/bin/bash
# default bench (as the driver runs it), then a two-rank rehearsal of the N > 1 job on the one GPU (gloo)
R=$PWD
O=$R/gpurun_out/${OUT:-r02e}
mkdir -p $O
timeout -k 10 600 python bench.py > $O/bench.json 2> $O/bench.err || { tail -20 $O/bench.err; exit 1; }
python tools/show.py $O/bench.json 2>/dev/null | head -60
RT355_BENCH_REHEARSAL=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 \
  bench.py --gpus 2 --steps 10 --warmup 2 --photons-per-gpu 200000 --no-other-configs --no-gather > $O/rehearsal.json 2> $O/rehearsal.err || { tail -30 $O/rehearsal.err; exit 1; }
cat $O/rehearsal.json
