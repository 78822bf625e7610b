#!/bin/bash
# same-box A/B: bash tools/exp_ab.sh <tag> "<env assignments for arm 1>" "<arm 2>" ...   (arm "base" = ppqsflhe_amd/libmkckks_base.so, a side build: make -C ppqsflhe_amd/csrc OUT=../libmkckks_base.so)
# every arm: bench.py --steps 20 --warmup 3 --no-cpu, arms interleaved twice
tag=$1; shift
out=gpurun_out/${tag}_ab.txt
: > $out
for rep in 1 2; do
  for arm in "$@"; do
    if [ "$arm" = "base" ]; then envs="MKCKKS_LIB=$PWD/ppqsflhe_amd/libmkckks_base.so"; else envs="$arm"; fi
    line=$(env $envs timeout -k 10 120 python bench.py --steps 20 --warmup 3 --no-cpu --min-seconds 1.5 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(round(d["value"]), round(d["ms_per_step"],4), round(d["ms_per_step_min"],4), round(d["ms_per_step_max"],4))')
    echo "[$arm] $line" | tee -a $out
  done
done
