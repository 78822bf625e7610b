#!/bin/bash
# r03 experiment: start-phase stagger of the first generation of workgroups (MKCKKS_STAGGER, percent of the built-in steps)
set -o pipefail
out=gpurun_out
export TMPDIR=/tmp
tools/probe_dispatch > $out/r03_probe_dispatch.txt 2>&1
for st in 0 50 100 200 0 100; do
    echo "stagger=$st $(MKCKKS_STAGGER=$st timeout -k 10 120 python bench.py --steps 20 --warmup 3 --no-cpu --min-seconds 1.5 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(round(d["value"]), d["ms_per_step"], d["ms_per_step_min"], d["ms_per_step_max"])')"
done | tee $out/r03_exp_stagger_bench.txt
for st in 0 100; do
    MKCKKS_STAGGER=$st timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/r03_stag${st}_trace -o runc -- python3 bench.py --steps 5 --warmup 2 --no-cpu --min-seconds 0 > $out/r03_stag${st}_trace_bench.json 2> $out/r03_stag${st}_trace.err
    python tools/kstats.py $out/r03_stag${st}_trace 7 > $out/r03_stag${st}_kernel_stats.txt
done
timeout -k 10 200 rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $out/r03_clk -o run -- python3 bench.py --steps 1 --warmup 0 --no-cpu --min-seconds 0 > $out/r03_clk.err 2>&1
echo done
