#!/bin/bash
# GPU box: the CLI parity tests, then the file-to-file measurement (-> gpurun_out/r03_server_round_e2e.txt)
set -o pipefail
out=gpurun_out
mkdir -p $out
timeout -k 10 300 python -m pytest tests/test_cli_hosts.py -m gpu -x -q > $out/cli_tests.log 2>&1; rc=$?
tail -5 $out/cli_tests.log
[ $rc -ne 0 ] && exit 1
timeout -k 10 600 python tools/e2e_server_round.py > $out/r03_server_round_e2e.txt 2> $out/r03_server_round_e2e.err; rc=$?
cut -c1-200 $out/r03_server_round_e2e.txt | grep -v "^    "; tail -5 $out/r03_server_round_e2e.err
exit $rc
