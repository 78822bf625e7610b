#!/bin/bash
# Reproduces the profile artefacts of a round on a one-GPU box:  bash tools/profile_round.sh <tag>
# -> gpurun_out/<tag>_*: kernel stats, HBM traffic (FETCH_SIZE, WRITE_SIZE in separate --pmc passes), achieved bytes / time
#    per kernel (with the algorithmic bytes beside the measured ones), SQ counters, bench lines; kernel_stats also carries
#    the per-step wall time (first kernel start -> last kernel end) next to the sum of kernel durations.  rocprofv3 always gets the program itself after `--`.  The merged n-client flow runs on one stream, so rocprof's
#    kernel durations add up to the step time.
set -e -o pipefail
tag=${1:-rXX}
out=gpurun_out
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_trace -o runc -- python3 bench.py --steps 5 --warmup 2 --no-cpu --max-blocks 1 > $out/${tag}_trace_bench.json 2> $out/${tag}_trace.err
python tools/kstats.py $out/${tag}_trace 7 > $out/${tag}_kernel_stats.txt
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/${tag}_fetch -o run -- python3 bench.py --steps 1 --warmup 0 --no-cpu --max-blocks 1 > $out/${tag}_pmc.err 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/${tag}_write -o run -- python3 bench.py --steps 1 --warmup 0 --no-cpu --max-blocks 1 >> $out/${tag}_pmc.err 2>&1
python tools/hbmtraffic.py $out/${tag}_fetch $out/${tag}_write 128 logn16_L12_dnum3_C8_B16 ${tag} > $out/${tag}_pmc_hbm.txt
cp profiles/hbm_traffic.json $out/hbm_traffic.json
python tools/kernel_bw.py $out/${tag}_fetch $out/${tag}_write $out/${tag}_trace 7 > $out/${tag}_kernel_bw.txt
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $out/${tag}_sq -o run -- python3 bench.py --steps 1 --warmup 0 --no-cpu --max-blocks 1 > $out/${tag}_sq.err 2>&1
python tools/sqsum.py $out/${tag}_sq > $out/${tag}_sq_counters.txt
python bench.py --steps 20 --warmup 3 > $out/${tag}_bench.json 2> $out/${tag}_bench.err
