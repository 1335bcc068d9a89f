# two-rank rehearsal of the bench's N > 1 control flow on one GPU (gloo; numbers mean nothing)
set -e
RT355_BENCH_REHEARSAL=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/rehearsal2.json 2> gpurun_out/rehearsal2.err
