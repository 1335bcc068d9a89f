set -e
timeout -k 10 200 python tools/exp_frame.py 2>/dev/null | tail -1
for v in prio3 prio6; do RT355_LIB=$PWD/gpurun_variants/librt355_$v.so timeout -k 10 200 python tools/exp_frame.py 2>/dev/null | tail -1; done
timeout -k 10 200 python tools/exp_frame.py 2>/dev/null | tail -1
