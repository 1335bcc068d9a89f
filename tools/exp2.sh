set -e
for v in cap256 cap320 cap384; do
  RT355_KNN_VERBOSE=1 RT355_LIB=$PWD/gpurun_variants/librt355_$v.so timeout -k 10 120 python bench.py --no-cpu-baseline --steps 5 > gpurun_out/exp_$v.json 2> gpurun_out/exp_$v.err
done
