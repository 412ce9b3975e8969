#!/bin/bash
# rocprofv3 kernel stats + HBM traffic of the prompt prefill alone (tools/bench_prefill.py 511), run on the GPU box
TAG=${1:-r2/prefill}
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
REPO=$PWD
cd /tmp
rm -rf /tmp/pp_stats /tmp/pp_pmc
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pp_stats -o stats -- python $REPO/tools/bench_prefill.py 511 > "$OUT/stats.log" 2>&1
f=$(find /tmp/pp_stats -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && grep -E "Name|lsk_" "$f" > "$OUT/kernel_stats_prefill.csv"
timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d /tmp/pp_pmc -o pmc -- python $REPO/tools/bench_prefill.py 511 > "$OUT/pmc.log" 2>&1
python $REPO/tools/pmc_summary.py /tmp/pp_pmc "$OUT/pmc_FETCH_SIZE.csv" >> "$OUT/pmc.log" 2>&1
tail -3 "$OUT/stats.log"
cat "$OUT/kernel_stats_prefill.csv" | cut -c1-150
grep -E "big|prefill|rmsnorm" "$OUT/pmc_FETCH_SIZE.csv" | cut -c1-200
