"""Aggregate rocprofv3 --pmc counter_collection CSVs per kernel: python tools/pmc_agg.py DIR [name-substring]."""
import csv
import glob
import sys
from collections import defaultdict

d = sys.argv[1]; sub = sys.argv[2] if len(sys.argv) > 2 else ''
agg = defaultdict(lambda: defaultdict(float)); cnt = defaultdict(lambda: defaultdict(int))
for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if sub not in k:
            continue
        k = k[:60]
        agg[k][r['Counter_Name']] += float(r['Counter_Value']); cnt[k][r['Counter_Name']] += 1
for k in agg:
    print(k)
    for c in sorted(agg[k]):
        print('   %-32s %16.0f  /launch (n=%d)' % (c, agg[k][c] / cnt[k][c], cnt[k][c]))
