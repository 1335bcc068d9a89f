#!/bin/bash
# GPU test run: the collected suite, then (if given) files pytest does not collect by itself
R=$PWD
O=$R/gpurun_out/${OUT:-r02b}
mkdir -p $O
timeout -k 10 1000 python -u -m pytest tests -m gpu -x -v --timeout 300 --timeout-method thread > $O/pytest.txt 2>&1
rc=$?
tail -5 $O/pytest.txt
[ $rc -ne 0 ] && exit $rc
for f in "$@"; do
  timeout -k 10 300 python -u -m pytest $f -x -v --timeout 200 --timeout-method thread > $O/$(basename $f).txt 2>&1 || { tail -30 $O/$(basename $f).txt; exit 1; }
  tail -3 $O/$(basename $f).txt
done
