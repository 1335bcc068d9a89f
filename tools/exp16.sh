# the Node host's GPU check
set -e
(cd raytracer_amd/host && timeout -k 10 300 node test/gpu_check.js) > gpurun_out/node_check.log 2>&1
