#!/bin/bash
set -o pipefail
A=$PWD/ppqsflhe_amd/libmkckks_aff.so
T2=$PWD/ppqsflhe_amd/libmkckks_tmaj2.so
MKCKKS_LIB=$T2 timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "modup_moddown or full_size" > gpurun_out/r03g_tests.log 2>&1; rc=$?
tail -2 gpurun_out/r03g_tests.log
[ $rc -ge 124 ] && exit 1
bash tools/exp_ab.sh r03g "MKCKKS_LIB=$A" "MKCKKS_LIB=$T2" "MKCKKS_LIB=$T2 MKCKKS_CU_AFFINE=1"
bash tools/exp_mempipe.sh r03g_aff0 MKCKKS_LIB=$A MKCKKS_CU_AFFINE=0
bash tools/exp_mempipe.sh r03g_aff1 MKCKKS_LIB=$A MKCKKS_CU_AFFINE=1
echo done
