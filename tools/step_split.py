"""Split one eager train step (tools/r6_list.sh -> step_list.txt: ordered launches with durations) into head / loss / optimizer / non-head and
the non-head part by kernel family.  Head = the split-layout (f16x3 / bf16x3) launches of the RetinaHead: conv_igemm_kernel<..., 3 | 2, 2, 0> on the
128- and 64-pixel tiles that follow to_split2, conv_wgrad_split_kernel, plus the head's unpack tail (the first backward_tail_kernel)."""
import re
import sys

rows = []
for ln in open(sys.argv[1]):
    m = re.match(r'\s*(\d+) (.+?)\s+wgs\s+(\d+)\s+([\d.]+)\s*$', ln)
    if m:
        rows.append((int(m.group(1)), m.group(2).strip(), int(m.group(3)), float(m.group(4))))
tot = sum(r[3] for r in rows)
first_split = next(i for i, r in enumerate(rows) if r[1].startswith('to_split2'))
head = loss = opt = 0.0
fam = {}
seen_tail = False
for i, (idx, name, wgs, us) in enumerate(rows):
    is_head = False
    if i > first_split and re.search(r'conv_igemm_kernel<float, (128|64), \d, \d, (3|2), 2, 0>', name) and wgs >= 1000:
        is_head = True
    if name.startswith('conv_wgrad_split_kernel'):
        is_head = True
    if name.startswith('backward_tail_kernel') and not seen_tail:
        seen_tail = True; is_head = True
    if is_head:
        head += us
    elif name.startswith('loss_') or (i > first_split and name.startswith(('at::native', '__amd_rocclr')) and not seen_tail):
        loss += us
    elif name.startswith('opt_') or name.startswith('prepare_params'):
        opt += us
    else:
        key = ('depthwise' if name.startswith('dw_') else 'squeeze-excite + gate multiply' if name.startswith(('se_', 'channel_scale')) else
               'BiFPN fusion' if name.startswith(('fuse_', 'to_split')) else 'backward tails' if name.startswith(('backward_tail', 'unpack')) else
               'weight gradients (thin / small)' if name.startswith('conv_wgrad') else
               'expand / project backward (fused kernels)' if name.startswith(('conv_pw_bwd', 'conv_pw_dgrad')) else
               'pointwise / BiFPN / stem convs' if name.startswith('conv_') else 'rest')
        fam[key] = fam.get(key, 0.0) + us
nonhead = sum(fam.values())
print('launches %d  total %.2f ms  head %.2f  loss %.2f  optimizer + parameter prep %.2f  non-head %.2f' % (len(rows), tot / 1e3, head / 1e3, loss / 1e3, opt / 1e3, nonhead / 1e3))
for k, v in sorted(fam.items(), key=lambda kv: -kv[1]):
    print('  %-45s %6.2f ms' % (k, v / 1e3))
