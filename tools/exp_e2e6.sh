#!/bin/bash
set -o pipefail
out=gpurun_out
mkdir -p $out
timeout -k 10 300 python -m pytest tests/test_cli_hosts.py -m gpu -x -q > $out/e2e_tests.log 2>&1; rc=$?
tail -5 $out/e2e_tests.log
[ $rc -ne 0 ] && exit 1
timeout -k 10 500 python tools/e2e_server_round.py --arms "MKCKKS_IO_THREADS=8;MKCKKS_IO_THREADS=8,MKCKKS_ROUND_CHUNK=1;MKCKKS_IO_THREADS=8,MKCKKS_ROUND_CHUNK=3;MKCKKS_IO_THREADS=8,MKCKKS_ROUND_CHUNK=6;MKCKKS_IO_THREADS=6;MKCKKS_IO_THREADS=12;MKCKKS_IO_THREADS=4;MKCKKS_IO_THREADS=2;MKCKKS_IO_THREADS=1;MKCKKS_SYNC_IO=1" > $out/r03_e2e_arms6.txt 2> $out/r03_e2e_arms6.err; rc=$?
cut -c1-420 $out/r03_e2e_arms6.txt; tail -5 $out/r03_e2e_arms6.err
exit $rc
