#!/bin/bash
# kNN gather: parity subset, then the per-kernel timing table
R=$PWD; O=$R/gpurun_out/r02i; mkdir -p $O
timeout -k 10 400 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread -k "knn or gather or no_alloc" > $O/tests.txt 2>&1 || { tail -30 $O/tests.txt; exit 1; }
tail -3 $O/tests.txt
RT355_KNN_VERBOSE=1 timeout -k 10 300 python -u tools/knn_bench.py > $O/knn_bench.txt 2>&1 || { tail -30 $O/knn_bench.txt; exit 1; }
grep -E "gathers|given up" $O/knn_bench.txt
