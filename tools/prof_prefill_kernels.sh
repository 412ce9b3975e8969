cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/s25   # (output directory: gpurun_out/s25)
cd $R
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/s25/prof -- python tools/prefill_yardstick.py --out gpurun_out/s25/y.json > gpurun_out/s25/log.txt 2>&1
python - <<'PY'
import csv,glob,collections
f=glob.glob('gpurun_out/s25/prof/**/*kernel_trace.csv',recursive=True)[0]
agg=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n=r['Kernel_Name']
    if 'lsk_' not in n: continue
    key=(n[:60], r['Grid_Size_X'] if 'Grid_Size_X' in r else r.get('Grid_Size',''), r.get('Grid_Size_Y',''))
    agg[key].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
out=open('gpurun_out/s25/kernels.txt','w')
for k,v in sorted(agg.items()):
    v=sorted(v); out.write(f"{k} n={len(v)} med={v[len(v)//2]:.2f} min={v[0]:.2f}\n")
PY
rm -rf gpurun_out/s25/prof
