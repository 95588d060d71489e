"""Per-step view of a rocprofv3 kernel trace (kt_kernel_trace.csv of `bench.py --no-graph`): the launches of the LAST step between
two optimizer kernels -- totals by duration bucket, the small-kernel table, and (with --list A B) the ordered launches A..B."""
import collections
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'opt_adamw' in r['Kernel_Name']]
step = rows[idx[-2] + 1:idx[-1] + 1]


def name(r):
    nm = re.sub(r'\(anonymous namespace\)::', '', r['Kernel_Name'])
    return re.sub(r'\(.*', '', nm).replace('void ', '')


def dur(r):
    return (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3


b, n = collections.Counter(), collections.Counter()
allk, alln = collections.Counter(), collections.Counter()
for r in step:
    d = dur(r)
    k = '<10' if d < 10 else '10-35' if d < 35 else '35-150' if d < 150 else '>150'
    b[k] += d; n[k] += 1
    allk[name(r)] += d; alln[name(r)] += 1
print('launches', len(step), 'sum of kernel time %.1f us' % sum(b.values()))
for k in ('<10', '10-35', '35-150', '>150'):
    print('  %-7s %4d launches %8.1f us' % (k, n[k], b[k]))
for k, v in allk.most_common(45):
    print('%-58s %4d %8.1f us  avg %6.1f' % (k[:58], alln[k], v, v / alln[k]))
if '--list' in sys.argv:
    a, e = int(sys.argv[sys.argv.index('--list') + 1]), int(sys.argv[sys.argv.index('--list') + 2])
    for i, r in enumerate(step):
        if a <= i <= e:
            print('%3d %-50s wgs %6d %7.1f' % (i, name(r)[:50], int(r['Grid_Size_X']) * int(r['Grid_Size_Y']) // int(r['Workgroup_Size_X']), dur(r)))
