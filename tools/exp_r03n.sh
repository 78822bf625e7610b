#!/bin/bash
set -o pipefail
out=gpurun_out
export TMPDIR=/tmp
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $out/r03n_tests.log 2>&1; rc=$?
tail -3 $out/r03n_tests.log
[ $rc -ge 124 ] && exit 1
bash tools/exp_ab.sh r03n "MKCKKS_LIB=$PWD/ppqsflhe_amd/libmkckks_ref2.so" "X=0"
echo done
