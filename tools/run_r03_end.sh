#!/bin/bash
# round-3 closing run on the GPU box: parity suite, profile artefacts, end-to-end serverRound, default bench
set -o pipefail
out=gpurun_out
mkdir -p $out
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $out/r03_end_tests.log 2>&1; rc=$?
tail -3 $out/r03_end_tests.log
[ $rc -ne 0 ] && exit 1
timeout -k 10 420 bash tools/profile_round.sh r03_end; rc=$?
[ $rc -ne 0 ] && { echo "profile_round rc=$rc"; tail -5 $out/r03_end_*.err; exit 1; }
cat $out/r03_end_bench.json
timeout -k 10 400 python tools/e2e_server_round.py > $out/r03_server_round_e2e.txt 2> $out/r03_server_round_e2e.err; rc=$?
cat $out/r03_server_round_e2e.txt; tail -5 $out/r03_server_round_e2e.err
[ $rc -ge 124 ] && exit 1
timeout -k 10 300 python bench.py > $out/r03_end_default_bench.json 2> $out/r03_end_default_bench.err; rc=$?
cat $out/r03_end_default_bench.json
exit $rc
