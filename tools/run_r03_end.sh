#!/bin/bash
# round-3 closing run on the GPU box: parity suite, smoke, profile artefacts, end-to-end serverRound, secondary shapes, default bench
set -o pipefail
out=gpurun_out
mkdir -p $out
timeout -k 10 700 python -m pytest tests -m gpu -x -q > $out/r03_end_tests.log 2>&1; rc=$?
tail -3 $out/r03_end_tests.log
[ $rc -ne 0 ] && exit 1
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $out/r03_end_smoke.log 2>&1; rc=$?
tail -2 $out/r03_end_smoke.log
[ $rc -ne 0 ] && exit 1
timeout -k 10 420 bash tools/profile_round.sh r03_end; rc=$?
[ $rc -ne 0 ] && { echo "profile_round rc=$rc"; tail -5 $out/r03_end_*.err; exit 1; }
cat $out/r03_end_bench.json
timeout -k 10 400 python tools/e2e_server_round.py > $out/r03_server_round_e2e.txt 2> $out/r03_server_round_e2e.err; rc=$?
cut -c1-200 $out/r03_server_round_e2e.txt | grep -v "^    "; tail -5 $out/r03_server_round_e2e.err
[ $rc -ge 124 ] && exit 1
timeout -k 10 200 python bench.py --log-n 17 --depth 18 --cts 8 --no-cpu > $out/r03_end_bench_n17.json 2> $out/r03_end_bench_n17.err; rc=$?
cut -c1-300 $out/r03_end_bench_n17.json
[ $rc -ge 124 ] && exit 1
timeout -k 10 200 python bench.py --log-n 14 --depth 2 --scaling-bits 40 --dnum 2 --cts 32 --no-cpu > $out/r03_end_bench_n14.json 2> $out/r03_end_bench_n14.err; rc=$?
cut -c1-300 $out/r03_end_bench_n14.json
[ $rc -ge 124 ] && exit 1
timeout -k 10 300 python bench.py > $out/r03_end_default_bench.json 2> $out/r03_end_default_bench.err; rc=$?
cut -c1-400 $out/r03_end_default_bench.json
exit $rc
