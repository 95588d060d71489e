R=$GRAFT_REPO_ROOT
for v in 1 0; do
(cd /tmp && EFFDET_FUSE_EXPAND_DW=$v timeout 200 rocprofv3 --kernel-trace --output-format csv -d $R/$OUT/kt_fuse$v -o kt -- python $R/tools/infer_bench.py --no-graph --reps 3 > $R/$OUT/kt_fuse$v.log 2>&1)
python - <<PY
import csv, glob, os
f = glob.glob('$OUT/kt_fuse$v/**/*kernel_trace.csv', recursive=True)
rows = sorted(csv.DictReader(open(f[0])), key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'nchw_to_nhwc' in r['Kernel_Name']]
i0 = idx[-1]
print('fuse=$v: first 40 launches of the last forward')
for r in rows[i0:i0 + 40]:
    n = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0]
    print('   %-52s %8.1f us  grid %s' % (n[:52], (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3, r['Grid_Size_X']))
PY
rm -rf $OUT/kt_fuse$v
done
