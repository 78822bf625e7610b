#!/bin/bash
set -o pipefail
out=gpurun_out
export TMPDIR=/tmp
for v in 0 2; do
  L=$PWD/ppqsflhe_amd/libmkckks_cl$v.so
  echo "== cl$v"
  MKCKKS_LIB=$L MKCKKS_CONV_LDS=1 timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/r03k_cl${v}_trace -o runc -- python3 bench.py --steps 5 --warmup 2 --no-cpu --min-seconds 0 > $out/r03k_cl${v}_bench.json 2> $out/r03k_cl${v}.err
  python tools/kstats.py $out/r03k_cl${v}_trace 7 | grep "k_conv_lds\|step 6"
done
echo done
