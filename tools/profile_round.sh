#!/bin/bash
# Round profile on the GPU box (run from the repo root through gpurun): rocprofv3 kernel stats of the bench command, then
# the PMC passes (separate runs, --kernel-trace only beside --pmc) of a short generation, summarised per kernel.
# Outputs (small) land in gpurun_out/$TAG/ ; copy what is to be judged into profiles/.
TAG=${1:-r02}
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
REPO=$PWD
cd /tmp
BENCH="python $REPO/bench.py --no-cpu-baseline --no-gpu-reference --no-sampled --no-operating-points --no-reference-parity --no-other-configs"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -o stats -- $BENCH --steps 2 --warmup 1 > "$OUT/stats_bench.log" 2>&1
f=$(find /tmp/prof_stats -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$OUT/kernel_stats_7B_spec.csv"
SHORT="$BENCH --weights gpu --steps 1 --warmup 0 --max-steps 48"
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES"; do
  tag=$(echo $pass | cut -d' ' -f1)
  rm -rf /tmp/prof_pmc
  timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc $pass -d /tmp/prof_pmc -o pmc -- $SHORT > "$OUT/pmc_$tag.log" 2>&1
  python $REPO/tools/pmc_summary.py /tmp/prof_pmc "$OUT/pmc_$tag.csv" >> "$OUT/pmc_$tag.log" 2>&1
done
python $REPO/tools/pmc_gateup_json.py "$OUT" "$OUT/pmc_gateup.json" "$TAG" >> "$OUT/pmc_FETCH_SIZE.log" 2>&1
ls -la "$OUT"
