# what the driver runs at the end of a round, in one go: GPU tests, smoke(), the Node host's check, the default bench
set -e
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 180 --timeout-method thread > gpurun_out/gpu_tests.log 2>&1
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/smoke.log 2>&1
(cd raytracer_amd/host && timeout -k 10 300 node test/gpu_check.js) > gpurun_out/node_check.log 2>&1
timeout -k 10 600 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
