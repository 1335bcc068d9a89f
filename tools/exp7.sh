# the photon-map leg of the bench (experiment)
set -e
timeout -k 10 300 python bench.py --no-cpu-baseline --no-other-configs > gpurun_out/exp_gather.json 2> gpurun_out/exp_gather.err
