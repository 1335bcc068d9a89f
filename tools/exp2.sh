set -e
export RT355_BENCH_NO_EVENTS=1
for v in 512_5 256_5 256_8; do
  RT355_LIB=$PWD/gpurun_variants/librt355_$v.so timeout -k 10 120 python bench.py --no-cpu-baseline --no-gather > gpurun_out/exp_v$v.json
done
timeout -k 10 120 python bench.py --no-cpu-baseline --no-gather > gpurun_out/exp_vbase.json
