#!/bin/bash
set -o pipefail
out=gpurun_out
mkdir -p $out
timeout -k 10 120 tools/ubench_hbm > $out/r03_ubench_hbm.txt 2>&1; rc=$?
cat $out/r03_ubench_hbm.txt
[ $rc -ne 0 ] && exit 1
export MKCKKS_LIB=$PWD/ppqsflhe_amd/libmkckks_sub.so
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "reencrypt_sum or config4" > $out/r03p_tests.log 2>&1; rc=$?
tail -3 $out/r03p_tests.log
[ $rc -ne 0 ] && exit 1
for rep in 1 2; do
for arm in X=0 MKCKKS_C1_SUB=16 MKCKKS_C1_SUB=32 MKCKKS_C1_SUB=8 MKCKKS_MODUP_SUB=16 MKCKKS_MODUP_SUB=32 MKCKKS_MODUP_SUB=64; do
  env $arm timeout -k 10 120 python bench.py --steps 10 --warmup 3 --no-cpu --max-blocks 12 > $out/r03p_b.json 2> $out/r03p_b.err; rc=$?
  [ $rc -ge 124 ] && exit 1
  python - "$arm" <<'PY' >> gpurun_out/r03p_ab.txt
import json,sys
try:
    j=json.loads(open('gpurun_out/r03p_b.json').read().strip().splitlines()[-1]); print(f"[{sys.argv[1]}] {j['value']:.0f} {j['ms_per_step']:.4f} {j['ms_per_step_min']:.4f} {j['ms_per_step_max']:.4f}")
except Exception as e: print(f"[{sys.argv[1]}] failed {e}")
PY
done; done
cat $out/r03p_ab.txt
