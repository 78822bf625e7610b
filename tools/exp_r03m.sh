#!/bin/bash
set -o pipefail
out=gpurun_out
export TMPDIR=/tmp
MKCKKS_CONV_LDS=1 timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "modup_moddown or reencrypt_sum or full_size or config5" > $out/r03m_tests.log 2>&1; rc=$?
tail -3 $out/r03m_tests.log
[ $rc -ge 124 ] && exit 1
bash tools/exp_ab.sh r03m "MKCKKS_LIB=$PWD/ppqsflhe_amd/libmkckks_ref2.so" "X=0" "MKCKKS_CONV_LDS=1"
MKCKKS_CONV_LDS=1 timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/r03m_trace -o runc -- python3 bench.py --steps 5 --warmup 2 --no-cpu --min-seconds 0 > $out/r03m_trace_bench.json 2> $out/r03m_trace.err
python tools/kstats.py $out/r03m_trace 7 > $out/r03m_kernel_stats.txt
grep "conv_lds\|step 6" $out/r03m_kernel_stats.txt
echo done
