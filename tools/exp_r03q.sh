#!/bin/bash
set -o pipefail
out=gpurun_out
mkdir -p $out
L=$PWD/ppqsflhe_amd
MKCKKS_LIB=$L/libmkckks_touch.so timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "reencrypt_sum or config4 or full_size or config5" > $out/r03q_tests.log 2>&1; rc=$?
tail -3 $out/r03q_tests.log
[ $rc -ne 0 ] && exit 1
bash tools/exp_ab.sh r03q "X=0" "MKCKKS_LIB=$L/libmkckks_touch.so" "X=0" "MKCKKS_LIB=$L/libmkckks_touch.so"
