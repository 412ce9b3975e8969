#!/bin/bash
# Kernel experiment driver (GPU box): prefill time + per-kernel stats of every variant build under layerskip_amd/csrc/variants/
TAG=${1:-r2/pfv}
ROWS=${2:-"511 2047"}
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
REPO=$PWD
cd /tmp
for lib in $REPO/layerskip_amd/csrc/variants/*.so; do
  v=$(basename $lib .so)
  export LSK_LIB=$lib
  timeout 200 python $REPO/tools/bench_prefill.py $ROWS > "$OUT/$v.log" 2>&1
  echo "== $v"; grep rows "$OUT/$v.log"
  [ -n "$NOPROF" ] && continue
  for n in $ROWS; do
    rm -rf /tmp/pfv_$v
    timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pfv_$v -o stats -- python $REPO/tools/bench_prefill.py $n > "$OUT/$v.$n.stats.log" 2>&1
    f=$(find /tmp/pfv_$v -name "*kernel_stats.csv" | head -1)
    [ -n "$f" ] && python - "$f" "$n" <<'PY' | tee "$OUT/$v.$n.kernels.txt"
import csv, sys
for r in csv.reader(open(sys.argv[1])):
    if any(k in r[0] for k in ("lsk_gemm_big", "prefill", "rmsnorm")):
        print(f"  rows {sys.argv[2]:>5} {r[0][:60]:60s} calls {r[1]:>4} avg_us {float(r[3]) / 1e3:8.2f}")
PY
  done
done
