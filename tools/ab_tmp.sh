set -e
timeout -k 10 300 python -u tools/knn_bench.py 2>&1 | grep -E "gathers/s" 
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread -k "knn or gather or no_alloc or config5" 2>&1 | tail -3
