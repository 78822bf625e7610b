#!/bin/bash
set -o pipefail
out=gpurun_out
export TMPDIR=/tmp
MKCKKS_LIB=$PWD/ppqsflhe_amd/libmkckks_stamp.so MKCKKS_STAMPS=1 timeout -k 10 300 python tools/stamps.py > $out/r03c_stamps.txt 2>&1; rc=$?
tail -30 $out/r03c_stamps.txt
[ $rc -ge 124 ] && exit 1
bash tools/exp_ab.sh r03c base "X=0" "MKCKKS_QSUM_PIPE=2"
timeout -k 10 400 python tools/exp_cumask.py > $out/r03c_exp_cumask.txt 2>&1; rc=$?
cat $out/r03c_exp_cumask.txt | tail -12
[ $rc -ge 124 ] && exit 1
timeout -k 10 900 python -m pytest tests/test_cli_hosts.py -m gpu -x -q > $out/r03c_cli_tests.log 2>&1
tail -5 $out/r03c_cli_tests.log
echo done
