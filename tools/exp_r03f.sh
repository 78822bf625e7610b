#!/bin/bash
set -o pipefail
out=gpurun_out
export TMPDIR=/tmp
A=$PWD/ppqsflhe_amd/libmkckks_aff.so
T=$PWD/ppqsflhe_amd/libmkckks_tmaj.so
MKCKKS_LIB=$T MKCKKS_CU_AFFINE=1 timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "modup_moddown or reencrypt_sum or full_size or rescale or ntt_roundtrip" > $out/r03f_tests.log 2>&1; rc=$?
tail -3 $out/r03f_tests.log
[ $rc -ge 124 ] && exit 1
bash tools/exp_ab.sh r03f "X=0" "MKCKKS_LIB=$A" "MKCKKS_LIB=$A MKCKKS_CU_AFFINE=1" "MKCKKS_LIB=$T" "MKCKKS_LIB=$T MKCKKS_CU_AFFINE=1"
rocprofv3 --list-avail > $out/r03_list_avail.txt 2>&1
for arm in aff1 tmaj; do
    if [ $arm = aff1 ]; then export MKCKKS_LIB=$A MKCKKS_CU_AFFINE=1; else export MKCKKS_LIB=$T MKCKKS_CU_AFFINE=0; fi
    timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/r03f_${arm}_trace -o runc -- python3 bench.py --steps 5 --warmup 2 --no-cpu --min-seconds 0 > $out/r03f_${arm}_trace_bench.json 2> $out/r03f_${arm}_trace.err
    python tools/kstats.py $out/r03f_${arm}_trace 7 > $out/r03f_${arm}_kernel_stats.txt
done
echo done
