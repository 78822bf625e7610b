#!/bin/bash
set -o pipefail
out=gpurun_out
mkdir -p $out
timeout -k 10 300 python -m pytest tests/test_cli_hosts.py -m gpu -x -q > $out/e2e_tests.log 2>&1; rc=$?
tail -5 $out/e2e_tests.log
[ $rc -ne 0 ] && exit 1
timeout -k 10 500 python tools/e2e_server_round.py --reps 3 --arms "MKCKKS_IO_THREADS=8" > $out/r03_e2e_arms9.txt 2> $out/r03_e2e_arms9.err; rc=$?
cut -c1-600 $out/r03_e2e_arms9.txt; tail -5 $out/r03_e2e_arms9.err
exit $rc
