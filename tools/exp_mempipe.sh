#!/bin/bash
# vector-memory pipeline counters of one bench step: bash tools/exp_mempipe.sh <tag> [env assignments...]
tag=$1; shift
out=gpurun_out
export TMPDIR=/tmp
i=0
for set in "GRBM_GUI_ACTIVE TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum" \
           "GRBM_GUI_ACTIVE TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum" \
           "GRBM_GUI_ACTIVE TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum" \
           "GRBM_GUI_ACTIVE TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum" \
           "GRBM_GUI_ACTIVE TCC_REQ_sum TCC_HIT_sum" \
           "GRBM_GUI_ACTIVE TCC_MISS_sum TCC_TAG_STALL_sum" \
           "GRBM_GUI_ACTIVE TCC_BUSY_avr"; do
    i=$((i+1))
    env "$@" timeout -k 10 200 rocprofv3 --pmc $set --output-format csv -d $out/${tag}_mp$i -o run -- python3 bench.py --steps 1 --warmup 0 --no-cpu --min-seconds 0 > $out/${tag}_mp$i.err 2>&1
done
python tools/mempipe.py $out/${tag}_mp1 $out/${tag}_mp2 $out/${tag}_mp3 $out/${tag}_mp4 $out/${tag}_mp5 $out/${tag}_mp6 $out/${tag}_mp7 > $out/${tag}_mempipe.txt
cat $out/${tag}_mempipe.txt
