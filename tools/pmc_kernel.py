"""Per-kernel means of every counter found under rocprofv3 --pmc output directories (+ mean dispatch duration from the
timestamps of the same pass):   python tools/pmc_kernel.py DIR [DIR ...] [--match SUBSTR]"""
import csv
import glob
import re
import sys
from collections import defaultdict

csv.field_size_limit(1 << 30)
dirs = [a for a in sys.argv[1:] if not a.startswith('--')]
match = None
if '--match' in sys.argv:
    match = sys.argv[sys.argv.index('--match') + 1]
    dirs = [d for d in dirs if d != match]


def short(name):
    m = re.search(r'(conv_igemm\w*_kernel<[^>(]*>|conv_wgrad\w*_kernel<[^>(]*>|[a-z_0-9]+_kernel)', name)
    return m.group(1).replace('unsigned short', 'bf16') if m else name[:60]


for d in dirs:
    acc = defaultdict(lambda: defaultdict(list)); dur = defaultdict(list)
    seen = set()
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r['Kernel_Name'])
            if match and match not in k:
                continue
            acc[k][r['Counter_Name']].append((r['Dispatch_Id'], float(r['Counter_Value'])))
            if r['Dispatch_Id'] not in seen:
                seen.add(r['Dispatch_Id']); dur[k].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
    print('==', d)
    for k, cs in acc.items():
        n = len(dur[k])
        print('%s  launches %d  mean duration %.1f us' % (k, n, sum(dur[k]) / max(n, 1)))
        for c, vals in sorted(cs.items()):
            per = defaultdict(float)
            for did, v in vals:
                per[did] += v
            print('    %-34s %16.0f' % (c, sum(per.values()) / len(per)))
