#!/bin/bash
set -o pipefail
out=gpurun_out
mkdir -p $out
L=$PWD/ppqsflhe_amd
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "config5 or n17 or c5s or reencrypt_sum" > $out/r03s_tests.log 2>&1; rc=$?
tail -2 $out/r03s_tests.log
[ $rc -ne 0 ] && exit 1
: > $out/r03s_ab.txt
for rep in 1 2; do
for arm in "MKCKKS_LIB=$L/libmkckks_ref3.so" "X=0"; do
  line=$(env $arm timeout -k 10 200 python bench.py --log-n 17 --depth 18 --cts 8 --no-cpu --min-seconds 1.5 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(round(d["value"]), round(d["ms_per_step"],4), round(d["ms_per_step_min"],4), round(d["ms_per_step_max"],4))')
  echo "[n17 $arm] $line" | tee -a $out/r03s_ab.txt
done; done
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/r03s_trace -o runc -- python3 bench.py --log-n 17 --depth 18 --cts 8 --steps 3 --warmup 1 --no-cpu --max-blocks 1 > $out/r03s_trace_bench.json 2> $out/r03s_trace.err
python tools/kstats.py $out/r03s_trace 4 > $out/r03s_n17_kernel_stats.txt
cat $out/r03s_n17_kernel_stats.txt
bash tools/exp_ab.sh r03s2 "MKCKKS_LIB=$L/libmkckks_ref3.so" "X=0"
