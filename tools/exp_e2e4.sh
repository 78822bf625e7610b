#!/bin/bash
set -o pipefail
out=gpurun_out
mkdir -p $out /dev/shm/mkprobe
timeout -k 10 200 tools/pcie_probe reads /dev/shm/mkprobe > $out/r03_read_probe.txt 2>&1; rc=$?
rm -rf /dev/shm/mkprobe
cat $out/r03_read_probe.txt
[ $rc -ne 0 ] && exit 1
nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null; taskset -p $$ | cut -c1-200; numactl -H 2>/dev/null | head -5
timeout -k 10 500 python tools/e2e_server_round.py --arms "MKCKKS_IO_THREADS=8,MKCKKS_ROUND_CHUNK=1,MKCKKS_IO_TRACE=$PWD/$out/iotrace_c1.txt;MKCKKS_IO_THREADS=8,MKCKKS_ROUND_CHUNK=2;MKCKKS_IO_THREADS=8,MKCKKS_ROUND_CHUNK=4;MKCKKS_IO_THREADS=12,MKCKKS_ROUND_CHUNK=2;MKCKKS_IO_THREADS=16,MKCKKS_ROUND_CHUNK=2;MKCKKS_IO_THREADS=4,MKCKKS_ROUND_CHUNK=2" > $out/r03_e2e_arms4.txt 2> $out/r03_e2e_arms4.err; rc=$?
cut -c1-420 $out/r03_e2e_arms4.txt; tail -5 $out/r03_e2e_arms4.err
exit $rc
