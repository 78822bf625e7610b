#!/bin/bash
set -o pipefail
out=gpurun_out
S=$PWD/ppqsflhe_amd/libmkckks_shuf.so
MKCKKS_LIB=$S timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "ntt_roundtrip or ntt_extreme or modup_moddown or full_size or rescale" > $out/r03i_tests.log 2>&1; rc=$?
tail -2 $out/r03i_tests.log
[ $rc -ge 124 ] && exit 1
bash tools/exp_ab.sh r03i "X=0" "MKCKKS_LIB=$S"
MKCKKS_LIB=$S timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/r03i_shuf_trace -o runc -- python3 bench.py --steps 5 --warmup 2 --no-cpu --min-seconds 0 > $out/r03i_shuf_trace_bench.json 2> $out/r03i_shuf_trace.err
python tools/kstats.py $out/r03i_shuf_trace 7 > $out/r03i_shuf_kernel_stats.txt
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/r03i_trace -o runc -- python3 bench.py --steps 5 --warmup 2 --no-cpu --min-seconds 0 > $out/r03i_trace_bench.json 2> $out/r03i_trace.err
python tools/kstats.py $out/r03i_trace 7 > $out/r03i_kernel_stats.txt
bash tools/exp_mempipe.sh r03i X=0
echo done
