set -e
for v in u4 u1 t256; do
  RT355_LIB=$PWD/gpurun_variants/librt355_$v.so timeout -k 10 120 python bench.py --no-cpu-baseline --no-gather > gpurun_out/exp_v$v.json
done
timeout -k 10 120 python bench.py --no-cpu-baseline --no-gather > gpurun_out/exp_vbase.json
