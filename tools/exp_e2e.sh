#!/bin/bash
set -o pipefail
out=gpurun_out
mkdir -p $out
timeout -k 10 300 python -m pytest tests/test_cli_hosts.py -m gpu -x -q > $out/e2e_tests.log 2>&1; rc=$?
tail -15 $out/e2e_tests.log
[ $rc -ne 0 ] && exit 1
timeout -k 10 500 python tools/e2e_server_round.py > $out/r03_server_round_e2e.txt 2> $out/r03_server_round_e2e.err; rc=$?
cat $out/r03_server_round_e2e.txt; tail -5 $out/r03_server_round_e2e.err
exit $rc
