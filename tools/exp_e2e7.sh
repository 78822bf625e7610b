#!/bin/bash
set -o pipefail
out=gpurun_out
mkdir -p $out
timeout -k 10 600 python tools/e2e_server_round.py > $out/r03_server_round_e2e.txt 2> $out/r03_server_round_e2e.err; rc=$?
cut -c1-330 $out/r03_server_round_e2e.txt; tail -5 $out/r03_server_round_e2e.err
exit $rc
