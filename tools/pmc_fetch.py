"""Average FETCH_SIZE (KiB, x2 on gfx950: MI355X_MICROARCH.md) and dispatch duration of the launches matching a kernel name in a
rocprofv3 --pmc FETCH_SIZE run.   python tools/pmc_fetch.py DIR MATCH [label]"""
import csv
import glob
import sys
from collections import defaultdict

csv.field_size_limit(1 << 30)
d, match = sys.argv[1], sys.argv[2]
label = sys.argv[3] if len(sys.argv) > 3 else ''
tot, dur = defaultdict(float), {}
for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if match in r['Kernel_Name'] and r['Counter_Name'] == 'FETCH_SIZE':
            tot[r['Dispatch_Id']] += float(r['Counter_Value'])
            dur[r['Dispatch_Id']] = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
if tot:
    n = len(tot)
    print('%-28s %-26s launches %3d  fetch %8.1f MB/launch  avg duration (profiled) %7.1f us' % (label, match, n, sum(tot.values()) * 2 * 1024 / n / 1e6, sum(dur.values()) / n))
else:
    print(label, match, 'no launches found')
