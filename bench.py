#!/usr/bin/env python
"""bench.py -- EfficientDet-D0 512px train step (fwd + focal/smooth-L1 loss + bwd + clip + AdamW) on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

Prints ONE JSON line on rank 0.  metric = BASELINE.json's "images/sec EfficientDet-D0 512px fwd+bwd";
workload = configs[2] (batch 32 per GPU @ 512x512, synthetic COCO-shape targets, random-init weights,
80 classes, W_bifpn 64 / D_bifpn 2).  Weak scaling: per-GPU batch fixed, value = total images / s.
Extra objects: roofline (dominant kernel = the bf16 MFMA implicit-GEMM conv, timed live with HIP events
on the launch stream), cpu_baseline (the oracle = torch-CPU restatement of the reference, bounded sample),
inference (configs[1]: batch-32 eval forward + decode + on-device NMS, ms/img).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

BF16_MFMA_PEAK_TFLOPS = 2500.0      # MI355X dense bf16 (MI355X_MICROARCH.md); fp32-input MFMA: 157.3
F32_MFMA_PEAK_TFLOPS = 157.3
TRAIN_GFLOP_PER_IMG = 192.15        # 3*64.089 - 0.113 (conv FLOPs, 2*MAC)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--batch', type=int, default=32, help='images per GPU')
    ap.add_argument('--size', type=int, default=512)
    ap.add_argument('--network', default='efficientdet-d0')
    ap.add_argument('--dtype', default='bf16', choices=['bf16', 'f32'])
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-inference', action='store_true')
    ap.add_argument('--no-roofline', action='store_true')
    ap.add_argument('--torch-optim', action='store_true', help='stock clip_grad_norm_ + torch.optim.AdamW(fused) instead of the HIP ClipAdamW')
    return ap.parse_args()


def cpu_baseline(network, size, seconds_budget=25.0):
    """The oracle (kind 'port': torch-CPU restatement of the reference, pinned on its golden vectors) timed on
    this host: B=4 forward + FocalLoss + backward at 512x512, as many repetitions as fit the budget."""
    from oracle import effdet_oracle as O
    nc, B = 80, 4
    cores = torch.get_num_threads()
    sd = O.make_state_dict(network, nc, seed=0)
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items() if v.is_floating_point() and 'running_' not in k
              and not k.startswith(('backbone._conv_head', 'backbone._bn1', 'backbone._fc'))}
    live = dict(sd); live.update(params)
    img, ann = O.synthetic_batch(B, size, seed=1, num_classes=nc)
    times = []
    t_end = time.perf_counter() + seconds_budget
    while True:
        t0 = time.perf_counter()
        cl, rl = O.train_losses(live, network, nc, img, ann)
        (cl.mean() + rl.mean()).backward()
        times.append(time.perf_counter() - t0)
        for p in params.values():
            p.grad = None
        if time.perf_counter() + times[-1] > t_end or len(times) >= 6:
            break
    best = min(times[1:]) if len(times) > 1 else times[0]
    return {'value': round(B / best, 3), 'unit': 'images/sec', 'cores': cores, 'kind': 'port',
            'sample': 'oracle (torch-CPU fp32 restatement of the reference) %s B=%d %dx%d fwd+loss+bwd, %d reps, best of reps after warm-up'
                      % (network, B, size, size, len(times))}


def main():
    a = parse()
    rank = int(os.environ.get('RANK', 0)); world = int(os.environ.get('WORLD_SIZE', 1)); local = int(os.environ.get('LOCAL_RANK', 0))
    assert world == a.gpus, 'launch with torch.distributed.run --nproc-per-node %d (WORLD_SIZE=%d)' % (a.gpus, world)
    ndev = torch.cuda.device_count()
    if local >= ndev:      # debug only (EFFDET_BENCH_BACKEND=gloo): several ranks sharing one GPU to exercise the N>1 control flow
        assert os.environ.get('EFFDET_BENCH_BACKEND') == 'gloo', 'rank %d has no GPU of its own (%d visible)' % (local, ndev)
        local %= ndev
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    from efficientdet.pytorch_amd import EfficientDet, EFFICIENTDET, ops, ddp
    import torch.distributed as dist
    if world > 1:
        ddp.init_process_group_from_env(os.environ.get('EFFDET_BENCH_BACKEND', 'nccl'))     # 'nccl' IS RCCL on ROCm
    dtype = torch.bfloat16 if a.dtype == 'bf16' else torch.float32
    cfg = EFFICIENTDET[a.network]
    torch.manual_seed(0)
    model = EfficientDet(num_classes=80, network=a.network, W_bifpn=cfg['W_bifpn'], D_bifpn=cfg['D_bifpn'],
                         D_class=cfg['D_class'], compute_dtype=dtype).to(dev)
    model.train(); model.is_training = True; model.freeze_bn()
    ddp.freeze_dead_parameters(model)
    net = ddp.wrap(model, device_ids=[local]) if world > 1 else model
    params = [p for p in model.parameters() if p.requires_grad]
    if a.torch_optim:
        opt = torch.optim.AdamW(params, lr=1e-4, fused=True)
    else:   # the same arithmetic (clip_grad_norm_(0.1) + AdamW(lr 1e-4, wd 1e-2)) as three HIP launches
        from efficientdet.pytorch_amd.optim import ClipAdamW
        opt = ClipAdamW(params, lr=1e-4, max_norm=0.1)

    # synthetic data resident in HBM before the timed region (randn images, COCO-shape targets)
    sys.path.insert(0, ROOT)
    from oracle.effdet_oracle import synthetic_batch     # input generator only (no compute): same seeded inputs as parity tests
    img, ann = synthetic_batch(a.batch, a.size, seed=1 + rank, num_classes=80)
    img, ann = img.to(dev), ann.to(dev)

    def step():
        opt.zero_grad(set_to_none=True)
        cl, rl = net([img, ann])
        loss = cl.mean() + rl.mean()
        loss.backward()
        if a.torch_optim:
            torch.nn.utils.clip_grad_norm_(params, 0.1)
        opt.step()
        return loss

    def sync_all():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        step()
    sync_all()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss = step()
    sync_all()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ms_step = dt / a.steps * 1e3
    value = a.batch * world * a.steps / dt

    out = {
        'metric': 'images/sec EfficientDet-D0 512px fwd+bwd', 'value': round(value, 2), 'unit': 'images/sec', 'n_gpus': world,
        'steps': a.steps, 'warmup': a.warmup, 'ms_per_step': round(ms_step, 3), 'higher_is_better': True, 'scaling': 'weak',
        'vs_baseline': None, 'dtype': a.dtype, 'data': 'synthetic',
        'config': {'workload': 'EfficientDet-D0 train step (fwd + FocalLoss/SmoothL1 + bwd + clip_grad_norm + AdamW), batch %d/GPU @ %dx%d, '
                               'synthetic COCO-shape targets, 80 classes, random-init, W_bifpn=%d D_bifpn=%d' % (a.batch, a.size, a.size, cfg['W_bifpn'], cfg['D_bifpn']),
                   'network': a.network, 'global_batch': a.batch * world, 'image_size': a.size, 'parallelism': 'dp%d' % world,
                   'final_loss': round(float(loss.item()), 4)},
        'algorithmic_tflops_per_gpu': round(TRAIN_GFLOP_PER_IMG * a.batch / ms_step, 2) if a.network == 'efficientdet-d0' and a.size == 512 else None,
    }

    if not a.no_roofline:
        # live per-launch timing of the MFMA kernels with HIP events on the launch stream (one instrumented step).
        # EVERY rank runs the step (under DDP its gradient all-reduce is collective); only rank 0 instruments it.
        if rank == 0:
            ops.PROFILE = ops.LaunchProfile()
        step()
        torch.cuda.synchronize()
    if rank == 0 and not a.no_roofline:
        summ = ops.PROFILE.summary(); ops.PROFILE = None
        peak = BF16_MFMA_PEAK_TFLOPS if a.dtype == 'bf16' else F32_MFMA_PEAK_TFLOPS
        hbm = {k: v for k, v in summ.items() if not k.startswith('conv_')}       # byte-counted (HBM-bound) kernels
        summ = {k: v for k, v in summ.items() if k.startswith('conv_')}          # flop-counted MFMA kernels
        dom = max(summ.items(), key=lambda kv: kv[1]['ms'])
        name, d = dom
        ach = d['flops'] / (d['ms'] * 1e-3) / 1e12
        # HBM bytes per launch of the dominant kernel: PMC counters cannot be read from inside this process; the value is
        # the one measured by `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (separate passes, FETCH x2 gfx950
        # correction) on this same command and committed under profiles/ (null when no such file / other dtype)
        traffic = None
        try:
            tj = json.load(open(os.path.join(ROOT, 'profiles', 'r01_hbm_traffic.json')))
            if a.dtype == 'bf16' and a.batch == 32 and a.size == 512:
                traffic = tj['kernels'].get(name, {}).get('hbm_bytes_per_launch')
        except Exception:
            traffic = None
        out['roofline'] = {'bound': 'mfma', 'kernel': name, 'achieved': round(ach, 2), 'peak': peak, 'unit': 'TFLOP/s',
                           'frac': round(ach / peak, 4), 'traffic': traffic,
                           'traffic_source': 'profiles/r01_hbm_traffic.json (rocprofv3 PMC, per launch)' if traffic else None,
                           'launches_per_step': d['launches'],
                           'avg_launch_ms': round(d['ms'] / d['launches'], 4),
                           'flops_per_launch': round(d['flops'] / d['launches'] / 1e9, 3),
                           'all_kernels': {k: {'launches': v['launches'], 'ms': round(v['ms'], 3),
                                               'tflops': round(v['flops'] / (v['ms'] * 1e-3) / 1e12, 2)} for k, v in summ.items()},
                           'hbm_kernels': {k: {'launches': v['launches'], 'ms': round(v['ms'], 3),
                                               'GBps': round(v['flops'] / (v['ms'] * 1e-3) / 1e9, 1)} for k, v in hbm.items()}}
    if world > 1:
        dist.barrier()

    if rank == 0 and world == 1 and not a.no_inference:
        # configs[1] is quoted on RANDOM-INIT weights (every anchor passes the 0.01 threshold = NMS worst case); the model
        # above has been updated by the timed AdamW steps, so use a fresh one
        del opt
        torch.manual_seed(0)
        model = EfficientDet(num_classes=80, network=a.network, W_bifpn=cfg['W_bifpn'], D_bifpn=cfg['D_bifpn'],
                             D_class=cfg['D_class'], is_training=False, compute_dtype=dtype).to(dev)
        model.eval(); model.is_training = False
        with torch.no_grad():
            for _ in range(2):
                model.detect(img)
            torch.cuda.synchronize(); t1 = time.perf_counter()
            reps = 5
            for _ in range(reps):
                dets = model.detect(img)
            torch.cuda.synchronize(); ti = (time.perf_counter() - t1) / reps
            for _ in range(2):
                model.forward_raw(img)
            torch.cuda.synchronize(); t1 = time.perf_counter()
            for _ in range(reps):
                model.forward_raw(img)
            torch.cuda.synchronize(); tf = (time.perf_counter() - t1) / reps
        out['inference'] = {'workload': 'D0 eval batch %d @ %d: forward + decode + per-image NMS (thr 0.01, IoU 0.5)' % (a.batch, a.size),
                            'ms_per_img': round(ti * 1e3 / a.batch, 4), 'forward_only_ms_per_img': round(tf * 1e3 / a.batch, 4),
                            'kept_boxes_img0': int(dets[0][0].numel())}

    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        out['cpu_baseline'] = cpu_baseline(a.network, a.size)

    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
