#!/bin/bash
set -o pipefail
out=gpurun_out
mkdir -p $out
timeout -k 10 500 python tools/e2e_server_round.py --arms "MKCKKS_IO_THREADS=8,MKCKKS_ROUND_CHUNK=1,MKCKKS_IO_TRACE=$PWD/$out/iotrace_c1.txt;MKCKKS_IO_THREADS=8,MKCKKS_ROUND_CHUNK=2,MKCKKS_IO_TRACE=$PWD/$out/iotrace_c2.txt;MKCKKS_IO_THREADS=8,MKCKKS_ROUND_CHUNK=4;MKCKKS_IO_THREADS=12,MKCKKS_ROUND_CHUNK=2;MKCKKS_IO_THREADS=16,MKCKKS_ROUND_CHUNK=2;MKCKKS_IO_THREADS=6,MKCKKS_ROUND_CHUNK=2;MKCKKS_IO_THREADS=8,MKCKKS_ROUND_CHUNK=3" > $out/r03_e2e_arms5.txt 2> $out/r03_e2e_arms5.err; rc=$?
cut -c1-420 $out/r03_e2e_arms5.txt; tail -5 $out/r03_e2e_arms5.err
grep "^#\|^c0\|^c1 " $out/iotrace_c1.txt $out/iotrace_c2.txt
exit $rc
