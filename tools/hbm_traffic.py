"""Aggregate `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes (separate runs of the same bench command) into
profiles/<round>_hbm_traffic.json: mean HBM bytes per launch of the main kernels.

    python tools/hbm_traffic.py FETCH_DIR WRITE_DIR OUT.json

FETCH_SIZE / WRITE_SIZE are reported in KiB; on gfx950 FETCH_SIZE tallies 128-byte requests at 64 B, so it is doubled
(MI355X_MICROARCH.md, HBM section).  WRITE_SIZE is uncalibrated (taken as is)."""
import csv
import glob
import json
import re
import sys
from collections import defaultdict


def per_launch(d, counter):
    tot = defaultdict(float)      # (kernel, dispatch) -> sum over counter instances
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if r['Counter_Name'] == counter:
                tot[(r['Kernel_Name'], r['Dispatch_Id'])] += float(r['Counter_Value'])
    agg = defaultdict(list)
    for (k, _), v in tot.items():
        agg[k].append(v)
    return agg


def short(name):
    m = re.search(r'conv_igemm_kernel<unsigned short, (\d+)', name)
    if m:
        return 'conv_igemm_kernel<bf16,%s>' % m.group(1)
    m = re.search(r'(conv_wgrad_tr_kernel<\d+>|conv_wgrad_kernel<unsigned short>|dw_\w+_lds_kernel|dw_wgrad_kernel|unpack_wgrad_kernel|'
                  r'loss_\w+_kernel|se_\w+_kernel|channel_scale_kernel|fuse_\w+_kernel)', name)
    return m.group(1).replace('unsigned short', 'bf16') if m else None


fetch, write = per_launch(sys.argv[1], 'FETCH_SIZE'), per_launch(sys.argv[2], 'WRITE_SIZE')
out = {}
for k in fetch:
    s = short(k)
    if s is None:
        continue
    e = out.setdefault(s, {'launches_sampled': 0, 'fetch': 0.0, 'write': 0.0})
    e['launches_sampled'] += len(fetch[k]); e['fetch'] += sum(fetch[k]) * 1024 * 2; e['write'] += sum(write.get(k, [])) * 1024
res = {'source': 'rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python bench.py --steps 2 --warmup 1; KiB -> bytes; '
                 'FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 counts 128-B requests as 64 B); WRITE_SIZE uncalibrated',
       'kernels': {k: {'launches_sampled': v['launches_sampled'],
                       'fetch_bytes_per_launch_x2_corrected': round(v['fetch'] / v['launches_sampled']),
                       'write_bytes_per_launch': round(v['write'] / v['launches_sampled']),
                       'hbm_bytes_per_launch': round((v['fetch'] + v['write']) / v['launches_sampled'])} for k, v in sorted(out.items())}}
json.dump(res, open(sys.argv[3], 'w'), indent=1)
print(json.dumps(res['kernels'].get('conv_igemm_kernel<bf16,128>'), indent=1))
