#!/bin/bash
set -o pipefail
out=gpurun_out
mkdir -p $out
timeout -k 10 400 python tools/e2e_server_round.py --reps 3 --arms "MKCKKS_IO_THREADS=8" > $out/r03_e2e_arms8.txt 2> $out/r03_e2e_arms8.err; rc=$?
cut -c1-900 $out/r03_e2e_arms8.txt; tail -5 $out/r03_e2e_arms8.err
exit $rc
