#!/bin/bash
set -o pipefail
out=gpurun_out
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > $out/r03l_tests.log 2>&1; rc=$?
tail -3 $out/r03l_tests.log
[ $rc -ge 124 ] && exit 1
bash tools/exp_ab.sh r03l base "X=0" "MKCKKS_CONV_LDS=1"
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/r03l_trace -o runc -- python3 bench.py --steps 5 --warmup 2 --no-cpu --min-seconds 0 > $out/r03l_trace_bench.json 2> $out/r03l_trace.err
python tools/kstats.py $out/r03l_trace 7 > $out/r03l_kernel_stats.txt
head -30 $out/r03l_kernel_stats.txt
echo done
