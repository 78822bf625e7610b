#!/bin/bash
set -o pipefail
out=gpurun_out
export TMPDIR=/tmp
tools/ubench_ops > $out/r03_ubench_ops.txt 2>&1
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "reencrypt_sum or unfused or full_size or config5 or chunks or deterministic or modup_moddown" > $out/r03b_tests.log 2>&1; rc=$?
tail -3 $out/r03b_tests.log
[ $rc -ge 124 ] && exit 1
bash tools/exp_ab.sh r03b base "X=0" "MKCKKS_QSUM_PIPE=1" "MKCKKS_QSUM_PIPE=2"
for arm in 0 1 2; do
    MKCKKS_QSUM_PIPE=$arm timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/r03b_pipe${arm}_trace -o runc -- python3 bench.py --steps 5 --warmup 2 --no-cpu --min-seconds 0 > $out/r03b_pipe${arm}_trace_bench.json 2> $out/r03b_pipe${arm}_trace.err
    python tools/kstats.py $out/r03b_pipe${arm}_trace 7 > $out/r03b_pipe${arm}_kernel_stats.txt
done
echo done
