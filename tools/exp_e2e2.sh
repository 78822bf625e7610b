#!/bin/bash
set -o pipefail
out=gpurun_out
mkdir -p $out
timeout -k 10 120 tools/pcie_probe > $out/r03_pcie_probe.txt 2>&1; rc=$?
cat $out/r03_pcie_probe.txt
[ $rc -ne 0 ] && exit 1
HSA_ENABLE_SDMA=0 timeout -k 10 120 tools/pcie_probe > $out/r03_pcie_probe_nosdma.txt 2>&1; rc=$?
cat $out/r03_pcie_probe_nosdma.txt
[ $rc -ne 0 ] && exit 1
B="MKCKKS_IO_THREADS=8,MKCKKS_ROUND_CHUNK=1"
timeout -k 10 500 python tools/e2e_server_round.py --arms "$B;$B,MKCKKS_UP_STREAMS=2;$B,MKCKKS_UP_STREAMS=3;$B,MKCKKS_PIN_FLAGS=0x80000000;$B,MKCKKS_PIN_FLAGS=0x20000000;$B,MKCKKS_UP_STREAMS=2,MKCKKS_PIN_FLAGS=0x80000000;$B,HSA_ENABLE_SDMA=0;MKCKKS_IO_THREADS=16,MKCKKS_ROUND_CHUNK=1,MKCKKS_UP_STREAMS=2;MKCKKS_IO_THREADS=4,MKCKKS_ROUND_CHUNK=1,MKCKKS_UP_STREAMS=2;MKCKKS_IO_THREADS=8,MKCKKS_ROUND_CHUNK=2,MKCKKS_UP_STREAMS=2" > $out/r03_e2e_arms.txt 2> $out/r03_e2e_arms.err; rc=$?
cat $out/r03_e2e_arms.txt; tail -5 $out/r03_e2e_arms.err
exit $rc
