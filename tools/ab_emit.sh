set -e
for v in old mb4 stream stream_mb4; do RT355_LIB=$PWD/gpurun_variants/librt355_$v.so timeout -k 10 200 python tools/exp_emit.py 2>/dev/null | tail -1; done
timeout -k 10 200 python tools/exp_emit.py 2>/dev/null | tail -1
