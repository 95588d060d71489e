"""Aggregate separate `rocprofv3 --pmc` passes of the same bench command into profiles/<round>_pmc.json:
per kernel symbol, mean HBM bytes per launch (FETCH_SIZE / WRITE_SIZE passes) and MFMA-pipe utilisation
(SQ_VALU_MFMA_BUSY_CYCLES / GRBM_GUI_ACTIVE pass).

    python tools/pmc_summary.py OUT.json DTYPE FETCH_DIR WRITE_DIR SQ_DIR

* FETCH_SIZE / WRITE_SIZE are reported in KiB; on gfx950 FETCH_SIZE tallies 128-byte requests at 64 B, so it is doubled
  (MI355X_MICROARCH.md, HBM section).  WRITE_SIZE is uncalibrated (taken as is).
* rocprofv3 sums a counter over its hardware instances: GRBM_GUI_ACTIVE arrives as the sum over the 8 XCDs, and
  SQ_VALU_MFMA_BUSY_CYCLES as the sum over all SIMDs (it counts cycles: 16 per v_mfma_f32_16x16x32_bf16).  So
      mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / ((GRBM_GUI_ACTIVE / 8) * 1024 SIMDs).
  The raw sums are kept next to the ratio so the normalisation can be re-checked."""
import csv
import glob
import json
import re
import sys
from collections import defaultdict

csv.field_size_limit(1 << 30)
NXCD, NSIMD = 8, 1024


DUR = defaultdict(list)       # kernel name -> dispatch durations (ns) of the SQ / GRBM pass, when the csv carries timestamps


def per_launch(d, counters):
    tot = defaultdict(float)
    seen = set()
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if r['Counter_Name'] in counters:
                tot[(r['Kernel_Name'], r['Dispatch_Id'], r['Counter_Name'])] += float(r['Counter_Value'])
                if 'GRBM_GUI_ACTIVE' in counters and r.get('Start_Timestamp') and (r['Kernel_Name'], r['Dispatch_Id']) not in seen:
                    seen.add((r['Kernel_Name'], r['Dispatch_Id']))
                    try:
                        DUR[r['Kernel_Name']].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
                    except (ValueError, KeyError):
                        pass
    agg = defaultdict(lambda: defaultdict(list))
    for (k, _, c), v in tot.items():
        agg[k][c].append(v)
    return agg


def short(name):
    m = re.search(r'conv_igemm_pers_kernel<(\d+), (\d+), (\d+)>', name)
    if m:
        return 'conv_igemm_pers_kernel<%s,%s,%s>' % m.groups()
    m = re.search(r'conv_igemm_kernel<(unsigned short|float), (\d+)(?:, \d+, \d+, (\d+))?', name)
    if m:
        if m.group(3) == '2':
            return 'conv_igemm_kernel<split,%s,bf16x3>' % m.group(2)
        if m.group(3) == '3':
            return 'conv_igemm_kernel<hsplit,%s,f16x3>' % m.group(2)
        return 'conv_igemm_kernel<%s,%s%s>' % ('bf16' if m.group(1) == 'unsigned short' else 'f32', m.group(2),
                                                ',bf16x3' if m.group(3) == '1' else '')
    if 'conv_wgrad_split_kernel' in name:
        return 'conv_wgrad_split_kernel'
    m = re.search(r'conv_wgrad_f32dma_kernel<(\d+), (\d+)>', name)
    if m:
        return 'conv_wgrad_f32dma_kernel<%s%s>' % (m.group(1), ',bf16x3' if m.group(2) == '1' else '')
    m = re.search(r'(conv_wgrad\w*_kernel<[^>(]*>|dw_\w+_kernel|unpack_wgrad\w*_kernel|wgrad_\w+_kernel|loss_\w+_kernel|se_\w+_kernel|'
                  r'channel_scale_kernel|fuse_\w+_kernel|opt_\w+_kernel|act_bwd_kernel|prepare_params_kernel|nms_\w+_kernel|decode_score_kernel|conv_pw_f32_kernel<\d+>|sort_\w+_kernel|radix_\w+_kernel|gather_dets_kernel|to_split_kernel|to_split2_kernel)', name)
    return m.group(1).replace('unsigned short', 'bf16').replace('float', 'f32') if m else None


def main():
    out_path, dtype, fdir, wdir, sdir = sys.argv[1:6]
    command = sys.argv[6] if len(sys.argv) > 6 else 'python bench.py --dtype %s --steps 2 --warmup 1 (train leg only)' % dtype
    fetch, write = per_launch(fdir, {'FETCH_SIZE'}), per_launch(wdir, {'WRITE_SIZE'})
    sq = per_launch(sdir, {'SQ_VALU_MFMA_BUSY_CYCLES', 'GRBM_GUI_ACTIVE', 'SQ_WAVE_CYCLES', 'SQ_BUSY_CYCLES'})
    out = {}
    for k, d in fetch.items():
        s = short(k)
        if s is None:
            continue
        e = out.setdefault(s, defaultdict(float))
        e['n_f'] += len(d['FETCH_SIZE']); e['fetch'] += sum(d['FETCH_SIZE']) * 1024 * 2
    for k, d in write.items():
        s = short(k)
        if s is None:
            continue
        e = out.setdefault(s, defaultdict(float))
        e['n_w'] += len(d['WRITE_SIZE']); e['write'] += sum(d['WRITE_SIZE']) * 1024
    for k, d in sq.items():
        s = short(k)
        if s is None:
            continue
        e = out.setdefault(s, defaultdict(float))
        e['n_s'] += len(d['GRBM_GUI_ACTIVE']); e['mfma'] += sum(d['SQ_VALU_MFMA_BUSY_CYCLES']); e['grbm'] += sum(d['GRBM_GUI_ACTIVE'])
        e['dur_ns'] += sum(DUR.get(k, [])); e['n_dur'] += len(DUR.get(k, []))
    kernels = {}
    for s, e in sorted(out.items()):
        r = {}
        if e['n_f'] and e['n_w']:
            r.update(launches_sampled=int(e['n_f']), fetch_bytes_per_launch_x2_corrected=round(e['fetch'] / e['n_f']),
                     write_bytes_per_launch=round(e['write'] / e['n_w']),
                     hbm_bytes_per_launch=round(e['fetch'] / e['n_f'] + e['write'] / e['n_w']))
        if e['n_s'] and e['grbm']:
            r.update(mfma_busy_cycles_per_launch=round(e['mfma'] / e['n_s']), grbm_gui_active_sum_per_launch=round(e['grbm'] / e['n_s']),
                     mfma_busy_frac=round(e['mfma'] / (e['grbm'] / NXCD * NSIMD), 4))
            if e['n_dur'] == e['n_s'] and e['dur_ns']:
                # effective shader clock while the kernel ran = GRBM_GUI_ACTIVE cycles (per XCD) / the dispatch's wall time (DVFS evidence)
                r.update(avg_duration_us_profiled=round(e['dur_ns'] / e['n_dur'] / 1e3, 1),
                         effective_clock_mhz=round((e['grbm'] / NXCD) / e['dur_ns'] * 1e3, 0))
        kernels[s] = r
    res = {'dtype': dtype,
           'source': 'rocprofv3 --pmc FETCH_SIZE | --pmc WRITE_SIZE | --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE (three separate passes) -- '
                     '%s; KiB -> bytes; FETCH_SIZE doubled per MI355X_MICROARCH.md; '
                     'mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs)' % command,
           'kernels': kernels}
    json.dump(res, open(out_path, 'w'), indent=1)
    for k in ('conv_igemm_kernel<hsplit,128,f16x3>', 'conv_igemm_kernel<bf16,128>', 'conv_igemm_kernel<f32,128>', 'conv_wgrad_tr_kernel<8>', 'conv_igemm_kernel<split,128,bf16x3>', 'conv_wgrad_split_kernel'):
        if k in kernels:
            print(k, json.dumps(kernels[k]))


if __name__ == '__main__':
    main()
