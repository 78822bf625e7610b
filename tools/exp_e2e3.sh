#!/bin/bash
set -o pipefail
out=gpurun_out
mkdir -p $out
B="MKCKKS_IO_THREADS=8,MKCKKS_ROUND_CHUNK=1"
timeout -k 10 500 python tools/e2e_server_round.py --arms "$B,MKCKKS_IO_TRACE=$PWD/$out/iotrace_c1.txt;MKCKKS_IO_THREADS=8,MKCKKS_ROUND_CHUNK=6,MKCKKS_IO_TRACE=$PWD/$out/iotrace_c6.txt" > $out/r03_e2e_arms3.txt 2> $out/r03_e2e_arms3.err; rc=$?
cat $out/r03_e2e_arms3.txt; tail -5 $out/r03_e2e_arms3.err
exit $rc
