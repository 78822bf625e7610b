#!/bin/bash
set -o pipefail
out=gpurun_out
mkdir -p $out
L=$PWD/ppqsflhe_amd
for v in 1 2; do
MKCKKS_Q0_SIDE=$v MKCKKS_LIB=$L/libmkckks_side.so timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "reencrypt_sum or config4 or full_size" > $out/r03r_tests$v.log 2>&1; rc=$?
tail -2 $out/r03r_tests$v.log
[ $rc -ne 0 ] && exit 1
done
bash tools/exp_ab.sh r03r "X=0" "MKCKKS_LIB=$L/libmkckks_side.so MKCKKS_Q0_SIDE=1" "MKCKKS_LIB=$L/libmkckks_side.so MKCKKS_Q0_SIDE=2"
