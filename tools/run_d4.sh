export TMPDIR=/tmp; mkdir -p /root/repo/gpurun_out/d4
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/d4/stats -o run -- python /root/repo/tools/infer_bench.py ${NET:---network efficientdet-d4 --batch 8 --size 1024} --reps 5 > /root/repo/gpurun_out/d4/log.txt 2>&1
cd /root/repo; tail -3 gpurun_out/d4/log.txt; find gpurun_out/d4 -name "*kernel_trace.csv" -delete
python - <<'PY'
import csv,re
rows=list(csv.DictReader(open('gpurun_out/d4/stats/run_kernel_stats.csv')))
tot=sum(float(r['TotalDurationNs']) for r in rows); print('total ms', tot/1e6)
for r in rows[:22]:
    n=r['Name']; m=re.search(r'([a-z_0-9]+_kernel(<[^>(]*>)?)',n); nm=m.group(1) if m else n[:50]
    print('%-56s calls %5s  ms %8.3f  avg us %8.1f  %5s%%'%(nm[:56], r['Calls'], float(r['TotalDurationNs'])/1e6, float(r['AverageNs'])/1e3, r['Percentage']))
PY
