"""Per-launch durations of the depthwise / squeeze-excite kernels from a rocprofv3 kernel trace (csv) of bench.py:
prints the LAST step's launches (name, grid, duration) so that each can be priced against its algorithmic bytes."""
import csv
import glob
import sys

pat = sys.argv[1]
keys = sys.argv[2].split(',') if len(sys.argv) > 2 else ['dw_', 'se_', 'channel_scale']
f = glob.glob(pat + '/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
sel = [r for r in rows if any(k in r['Kernel_Name'] for k in keys)]
# one step = launches between two consecutive opt_adamw kernels
marks = [i for i, r in enumerate(rows) if 'opt_adamw' in r['Kernel_Name']]
lo, hi = (marks[-2], marks[-1]) if len(marks) >= 2 else (0, len(rows))
for r in rows[lo:hi]:
    if any(k in r['Kernel_Name'] for k in keys):
        n = r['Kernel_Name'].replace('void (anonymous namespace)::', '').split('(')[0]
        print('%-52s grid %6s x %4s  wg %4s  %8.1f us' % (n[:52], r['Grid_Size_X'], r['Grid_Size_Y'], r['Workgroup_Size_X'],
                                                          (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3))
