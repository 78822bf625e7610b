#!/bin/bash
set -o pipefail
out=gpurun_out
export TMPDIR=/tmp
for v in "" _p1 _p2; do
  MKCKKS_LIB=$PWD/ppqsflhe_amd/libmkckks_stamp$v.so MKCKKS_STAMPS=1 timeout -k 10 300 python tools/stamps.py 2>&1 | grep -v amdgpu.ids > $out/r03h_stamps$v.txt; rc=$?
  [ $rc -ge 124 ] && exit 1
  echo "== stamps$v"; head -14 $out/r03h_stamps$v.txt
done
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "unfused or modup_moddown or reencrypt_sum" > $out/r03h_tests.log 2>&1; rc=$?
tail -3 $out/r03h_tests.log
[ $rc -ge 124 ] && exit 1
bash tools/exp_ab.sh r03h base "X=0" "MKCKKS_CONV_PAIR2=1" "MKCKKS_CONV_PAIR2=2" "MKCKKS_CONV_PAIR2=3"
echo done
