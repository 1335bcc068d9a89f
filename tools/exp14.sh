# register caps of the first pass x chunks per workgroup (experiment)
set -e
for v in w7 w6 base; do
  for c in 1 4; do
    L=""; if [ $v != base ]; then L=$PWD/gpurun_variants/librt355_$v.so; fi
    RT355_LIB=$L RT355_PRIMARY_CHUNKS=$c RT355_BENCH_SERIAL_FRAMES=1 timeout -k 10 120 python bench.py --no-cpu-baseline --no-gather --no-other-configs --steps 40 > gpurun_out/exp_${v}_c${c}_serial.json
    RT355_LIB=$L RT355_PRIMARY_CHUNKS=$c timeout -k 10 120 python bench.py --no-cpu-baseline --no-gather --no-other-configs --steps 40 > gpurun_out/exp_${v}_c$c.json
  done
done
