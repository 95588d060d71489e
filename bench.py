#!/usr/bin/env python
"""bench.py -- EfficientDet-D0 512px train step (fwd + focal/smooth-L1 loss + bwd + clip + AdamW) on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

Prints ONE JSON line on rank 0.  metric = BASELINE.json's "images/sec EfficientDet-D0 512px fwd+bwd";
workload = configs[2] (batch 32 per GPU @ 512x512, synthetic COCO-shape targets, random-init weights,
80 classes, W_bifpn 64 / D_bifpn 2).  Weak scaling: per-GPU batch fixed, value = total images / s.

The HEADLINE (top-level value / dtype / roofline) is the fastest arithmetic mode whose `-m gpu` tests hold north_star's 1e-3
gates against the real reference's goldens: fp32 storage with bf16x3 MFMA products (`--dtype f32_bf16x3`, the default).
Extra objects: roofline (dominant MFMA kernel, timed live with HIP events on the launch stream), strict_mode_f32 (the SAME
step with exact-fp32 MFMA products), throughput_mode_bf16 (bf16 storage + bf16 products: 2.5e-2 gates, NOT a parity mode),
inference (configs[1]: batch-32 eval forward + decode + on-device NMS, ms/img, all modes), inference_d4 (configs[4]: D4
batch 8 @ 1024), cpu_baseline (the oracle = torch-CPU restatement of the reference, bounded sample, thread sweep +
configs[0]).  Every train leg is timed over >= 20 steps.
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
    ap.add_argument('--dtype', default='f32_bf16x3', choices=['bf16', 'f32', 'f32_bf16x3'],
                    help='arithmetic mode of the headline leg (default: the fastest mode that meets the 1e-3 parity gates)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-inference', action='store_true')
    ap.add_argument('--no-roofline', action='store_true')
    ap.add_argument('--no-graph', action='store_true', help='eager launches instead of replaying the captured hipGraph of the step (N=1)')
    ap.add_argument('--no-extra-modes', '--no-parity-mode', dest='no_extra_modes', action='store_true',
                    help='skip the legs of the other two arithmetic modes')
    ap.add_argument('--no-d4', action='store_true', help='skip configs[4] (D4 batch 8 @ 1024 inference)')
    ap.add_argument('--extra-steps', type=int, default=20, help='timed steps of each extra-mode train leg')
    ap.add_argument('--extra-warmup', type=int, default=3)
    ap.add_argument('--no-ddp-graph', action='store_true',
                    help='N > 1: eager launches under DDP.  Default: the DDP step, RCCL all-reduces included, is captured as ONE hipGraph '
                         'when a pre-flight probe (throw-away child process per rank: world_size-1 RCCL group, all-reduce captured + '
                         'replayed) says this torch / RCCL build can do it on this box; every rank must agree, else all run eager')
    ap.add_argument('--ddp-graph', action='store_true', help='(kept for older command lines: the captured DDP step is the default now)')
    ap.add_argument('--ddp-single', action='store_true',
                    help='N = 1 through the N > 1 code path: world_size-1 RCCL process group, ddp.wrap, bucketed all-reduce, captured DDP step '
                         '(what a box with one GPU can validate of the multi-GPU path)')
    ap.add_argument('--infer-reps', type=int, default=20, help='timed repetitions of every inference leg')
    ap.add_argument('--torch-optim', action='store_true', help='stock clip_grad_norm_ + torch.optim.AdamW(fused) instead of the HIP ClipAdamW')
    return ap.parse_args()


def cpu_baseline(network, size, seconds_budget=25.0):
    """The oracle (kind 'port': torch-CPU restatement of the reference, pinned on its golden vectors) timed on this host:
    forward + FocalLoss + backward at 512x512 over a sweep of intra-op thread counts (an over-subscribed pool is SLOWER than
    a moderate one for these small-batch convs), best reported with its thread count; plus BASELINE configs[0]
    (B=1 forward to (cls, reg, anchors))."""
    from oracle import effdet_oracle as O
    nc, B = 80, 2
    ncpu = os.cpu_count() or 1
    sd = O.make_state_dict(network, nc, seed=0)
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items() if v.is_floating_point() and 'running_' not in k
              and not k.startswith(('backbone._conv_head', 'backbone._bn1', 'backbone._fc'))}
    live = dict(sd); live.update(params)
    img, ann = O.synthetic_batch(B, size, seed=1, num_classes=nc)

    def one():
        t0 = time.perf_counter()
        cl, rl = O.train_losses(live, network, nc, img, ann)
        (cl.mean() + rl.mean()).backward()
        dt = time.perf_counter() - t0
        for p in params.values():
            p.grad = None
        return dt
    t_start = time.perf_counter()
    sweep, first = {}, True
    for nt in [t for t in (8, 16, 32, 64, 128) if t <= ncpu] or [ncpu]:
        torch.set_num_threads(nt)
        if first:
            one(); first = False                       # warm-up (allocator, oneDNN primitive cache)
        ts = [one()]
        if time.perf_counter() - t_start < seconds_budget * 0.7:
            ts.append(one())
        sweep[nt] = round(B / min(ts), 3)
        if time.perf_counter() - t_start > seconds_budget:
            break
    best_nt = max(sweep, key=sweep.get)
    torch.set_num_threads(best_nt)
    with torch.no_grad():                               # configs[0]: D0, 1x3x512x512, forward only
        O.forward_raw(sd, network, nc, img[:1])
        f = []
        for _ in range(3):
            t0 = time.perf_counter(); O.forward_raw(sd, network, nc, img[:1]); f.append(time.perf_counter() - t0)
    return {'value': sweep[best_nt], 'unit': 'images/sec', 'cores': best_nt, 'kind': 'port', 'host_cpus': ncpu,
            'threads_sweep_img_per_s': sweep,
            'config0_forward_ms': round(min(f) * 1e3, 1), 'config0_threads': best_nt,
            'sample': 'oracle (torch-CPU fp32 restatement of the reference) %s B=%d %dx%d fwd+loss+bwd per thread count (best of <=2 reps after '
                      'one warm-up); config0 = B=1 forward, best of 3' % (network, B, size, size)}


def build_model(network, dtype, dev, training, f32_arith='f32'):
    from efficientdet.pytorch_amd import EfficientDet, EFFICIENTDET
    cfg = EFFICIENTDET[network]
    torch.manual_seed(0)
    m = EfficientDet(num_classes=80, network=network, W_bifpn=cfg['W_bifpn'], D_bifpn=cfg['D_bifpn'], D_class=cfg['D_class'],
                     is_training=training, compute_dtype=dtype, f32_arith=f32_arith).to(dev)
    if training:
        m.train(); m.is_training = True; m.freeze_bn()
    else:
        m.eval(); m.is_training = False
    return m


def roofline_of(summ, dtype_name, batch, size):
    """Dominant MFMA kernel of one instrumented step (per-launch HIP events on the launch stream) against the dense peak."""
    # bf16x3: every algorithmic MAC costs three bf16 MFMA MACs -> the dense bf16 peak / 3 in algorithmic FLOP/s
    peak = {'bf16': BF16_MFMA_PEAK_TFLOPS, 'f32': F32_MFMA_PEAK_TFLOPS, 'f32_bf16x3': round(BF16_MFMA_PEAK_TFLOPS / 3.0, 1)}[dtype_name]
    hbm = {k: v for k, v in summ.items() if not k.startswith('conv_')}       # byte-counted (HBM-bound) kernels
    mf = {k: v for k, v in summ.items() if k.startswith('conv_')}            # flop-counted MFMA kernels
    name, d = max(mf.items(), key=lambda kv: kv[1]['ms'])
    ach = d['flops'] / (d['ms'] * 1e-3) / 1e12
    # HBM bytes per launch and MFMA-pipe utilisation of the dominant kernel: PMC counters cannot be read from inside this
    # process; the values are those measured by separate `rocprofv3 --pmc` passes on this same command and committed
    # under profiles/ (null when no such file / another dtype or shape)
    traffic = util = src = None
    import glob
    for fn in sorted((os.path.basename(f) for f in glob.glob(os.path.join(ROOT, 'profiles', 'r*_pmc*.json'))), reverse=True) + ['r01_hbm_traffic.json']:
        try:
            tj = json.load(open(os.path.join(ROOT, 'profiles', fn)))
            if tj.get('dtype', 'bf16') == dtype_name and batch == 32 and size == 512:
                kk = tj['kernels'].get(name, {})
                traffic, util, src = kk.get('hbm_bytes_per_launch'), kk.get('mfma_busy_frac'), 'profiles/' + fn
                break
        except Exception:
            pass
    return {'bound': 'mfma', 'kernel': name, 'achieved': round(ach, 2), 'peak': peak, 'unit': 'TFLOP/s', 'frac': round(ach / peak, 4),
            'traffic': traffic, 'mfma_busy_frac': util, 'traffic_source': (src + ' (rocprofv3 PMC, per launch)') if src else None,
            'launches_per_step': d['launches'], 'avg_launch_ms': round(d['ms'] / d['launches'], 4),
            'flops_per_launch': round(d['flops'] / d['launches'] / 1e9, 3),
            'all_kernels': {k: {'launches': v['launches'], 'ms': round(v['ms'], 3), 'tflops': round(v['flops'] / (v['ms'] * 1e-3) / 1e12, 2)}
                            for k, v in mf.items()},
            'hbm_kernels': {k: {'launches': v['launches'], 'ms': round(v['ms'], 3), 'GBps': round(v['flops'] / (v['ms'] * 1e-3) / 1e9, 1)}
                            for k, v in hbm.items()}}


def train_leg(a, dtype_name, steps, warmup, rank, world, local, dev, want_roofline):
    """EXACTLY `steps` timed train steps (after `warmup` untimed ones) in one compute dtype -> (img/s, ms/step, loss, roofline)."""
    import torch.distributed as dist
    from efficientdet.pytorch_amd import ops, ddp
    from efficientdet.pytorch_amd.optim import ClipAdamW
    from efficientdet.pytorch_amd.synthetic import synthetic_batch      # the package's own generator: the GPU legs are oracle-free
    dtype = torch.bfloat16 if dtype_name == 'bf16' else torch.float32
    model = build_model(a.network, dtype, dev, True, 'bf16x3' if dtype_name == 'f32_bf16x3' else 'f32')
    ddp.freeze_dead_parameters(model)
    use_ddp = world > 1 or a.ddp_single
    # (captured DDP step: the module lives on the stream GraphedTrainStep warms up and captures on; eager DDP: the ambient stream)
    want_graph = use_ddp and a.ddp_graph_ok and not a.no_graph and not a.torch_optim
    net = (ddp.wrap_for_capture(model, device_ids=[local]) if want_graph else ddp.wrap(model, device_ids=[local])) if use_ddp else model
    params = [p for p in model.parameters() if p.requires_grad]
    if a.torch_optim:
        opt = torch.optim.AdamW(params, lr=1e-4, fused=True)
    else:   # the same arithmetic (clip_grad_norm_(0.1) + AdamW(lr 1e-4, wd 1e-2)) as three HIP launches
        opt = ClipAdamW(params, lr=1e-4, max_norm=0.1)
    # synthetic data resident in HBM before the timed region (randn images, COCO-shape targets)
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
        if use_ddp:
            dist.barrier()
        torch.cuda.synchronize()

    import contextlib
    cap_stream = getattr(net, '_effdet_capture_stream', None)      # eager steps of a capture-bound DDP module run on ITS stream too

    def on_stream():
        return torch.cuda.stream(cap_stream) if cap_stream is not None else contextlib.nullcontext()
    with on_stream():
        for _ in range(warmup):
            step()
    sync_all()
    graphed = None
    if (not use_ddp or a.ddp_graph_ok) and not a.no_graph and not a.torch_optim:
        # the SAME step (zero_grad, forward, loss, backward, clip + AdamW) captured once as a hipGraph and replayed: one
        # hipGraphLaunch per step instead of ~400 launches through Python; every replay does the full work on the resident
        # batch (fresh drop_connect masks from the device-side step counter, parameters updated in place).  Under DDP the bucketed
        # RCCL all-reduces are captured with the step (11 eager iterations first: DDP rebuilds its buckets after the first one);
        # whether RCCL can be captured here was decided for ALL ranks by the pre-flight probe in main(), before any collective
        # of this job existed -- a capture that fails now, with collectives in flight, is not recoverable and aborts the run.
        from efficientdet.pytorch_amd.graph import GraphedTrainStep
        try:
            graphed = GraphedTrainStep(net, opt, img, ann, warmup=11 if use_ddp else 2)
            for _ in range(2):
                graphed()
            sync_all()
        except Exception as e:        # report and fall back to eager launches
            if use_ddp:
                sys.stderr.write('capturing the DDP step failed AFTER the RCCL capture probe passed (%s: %s); re-run with --no-ddp-graph\n'
                                 % (type(e).__name__, e))
                raise
            sys.stderr.write('hipGraph capture failed (%s: %s); timing eager launches\n' % (type(e).__name__, e))
            graphed = None
    sync_all()
    t0 = time.perf_counter()
    if graphed is not None:
        for _ in range(steps):
            cl_rl = graphed()
        loss = cl_rl[0].mean() + cl_rl[1].mean()
    else:
        with on_stream():
            for _ in range(steps):
                loss = step()
    host_dt = time.perf_counter() - t0          # time the HOST needed to issue the steps (a host-bound regime shows as host ~= wall)
    sync_all()
    dt = time.perf_counter() - t0
    host_ms = [round(host_dt / steps * 1e3, 3)]
    if use_ddp:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        hm = [torch.zeros(1, device=dev, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(hm, torch.tensor([host_ms[0]], device=dev, dtype=torch.float64))
        host_ms = [round(float(x.item()), 3) for x in hm]
    roof = None
    if want_roofline:
        # live per-launch timing with HIP events on the launch stream (one instrumented step).  EVERY rank runs the step
        # (under DDP its gradient all-reduce is collective); only rank 0 instruments it.
        if rank == 0:
            ops.PROFILE = ops.LaunchProfile()
        with on_stream():
            step()
        torch.cuda.synchronize()
        if rank == 0:
            summ = ops.PROFILE.summary()
            roof = roofline_of(summ, dtype_name, a.batch, a.size)
            # the dominant symbol's launches by shape: its MFMA-bound head shapes apart from the HBM-bound backbone ones
            roof['dominant_by_shape'] = ops.PROFILE.by_shape(roof['kernel'])
            roof['other_mfma_by_shape'] = {k: ops.PROFILE.by_shape(k, top=3) for k, v in summ.items()
                                           if k.startswith('conv_') and k != roof['kernel'] and v['ms'] >= 1.0}
            ops.PROFILE = None
        if use_ddp:
            dist.barrier()
    final = float(loss.item())
    del opt, net, model
    torch.cuda.empty_cache()
    return a.batch * world * steps / dt, dt / steps * 1e3, final, roof, img, graphed is not None, host_ms


INFER_GFLOP_PER_IMG = {('efficientdet-d0', 512): 64.089, ('efficientdet-d4', 1024): 455.596}     # SURVEY 8(d) forward conv FLOPs (2*MAC)


def inference_roofline(model, img, dtype_name):
    """The dominant MFMA kernel of ONE instrumented eager forward (per-launch HIP events on the launch stream) against the dense
    peak of its arithmetic, like the train legs' roofline; traffic = null (no PMC pass of the inference command is attached)."""
    from efficientdet.pytorch_amd import ops
    peak = {'bf16': BF16_MFMA_PEAK_TFLOPS, 'f32': F32_MFMA_PEAK_TFLOPS, 'f32_bf16x3': round(BF16_MFMA_PEAK_TFLOPS / 3.0, 1)}[dtype_name]
    ops.PROFILE = ops.LaunchProfile()
    try:
        with torch.no_grad():
            model.forward_raw(img)
        torch.cuda.synchronize()
        summ = ops.PROFILE.summary()
        mf = {k: v for k, v in summ.items() if k.startswith('conv_')}
        name, d = max(mf.items(), key=lambda kv: kv[1]['ms'])
        ach = d['flops'] / (d['ms'] * 1e-3) / 1e12
        roof = {'bound': 'mfma', 'kernel': name, 'achieved': round(ach, 2), 'peak': peak, 'unit': 'TFLOP/s', 'frac': round(ach / peak, 4),
                'traffic': None, 'launches_per_forward': d['launches'], 'avg_launch_ms': round(d['ms'] / d['launches'], 4),
                'flops_per_launch': round(d['flops'] / d['launches'] / 1e9, 3),
                'dominant_by_shape': ops.PROFILE.by_shape(name, top=3),
                'all_kernels': {k: {'launches': v['launches'], 'ms': round(v['ms'], 3), 'tflops': round(v['flops'] / (v['ms'] * 1e-3) / 1e12, 2)}
                                for k, v in mf.items()},
                'hbm_kernels': {k: {'launches': v['launches'], 'ms': round(v['ms'], 3), 'GBps': round(v['flops'] / (v['ms'] * 1e-3) / 1e9, 1)}
                                for k, v in summ.items() if not k.startswith('conv_')}}
    finally:
        ops.PROFILE = None
    return roof


def inference_leg(network, dtype, dev, img, reps=20, graph=True, f32_arith='f32', dtype_name=None):
    """eval forward + decode + per-image NMS (thr 0.01, IoU 0.5) on RANDOM-INIT weights: every anchor passes the threshold = the
    NMS worst case.  -> (ms/img end to end, ms/img forward only, kept boxes of image 0, roofline of the forward or None)."""
    model = build_model(network, dtype, dev, False, f32_arith)
    B = img.shape[0]
    with torch.no_grad():
        for _ in range(2):
            model.detect(img)
        detect = lambda: model.detect(img)
        if graph:      # the same forward + decode + NMS + gather replayed as ONE hipGraph
            from efficientdet.pytorch_amd.graph import GraphedDetect
            try:
                gd = GraphedDetect(model, img)
                gd(); detect = gd
            except Exception as e:
                sys.stderr.write('hipGraph capture of detect failed (%s: %s); timing eager launches\n' % (type(e).__name__, e))
                graph = False
        torch.cuda.synchronize(); t1 = time.perf_counter()
        for _ in range(reps):
            dets = detect()
        torch.cuda.synchronize(); ti = (time.perf_counter() - t1) / reps
        for _ in range(2):
            model.forward_raw(img)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        for _ in range(reps):
            model.forward_raw(img)
        torch.cuda.synchronize(); tf = (time.perf_counter() - t1) / reps
    kept = int(dets[0][0].numel())
    roof = inference_roofline(model, img, dtype_name) if dtype_name else None
    del model, detect, dets
    torch.cuda.empty_cache()
    return round(ti * 1e3 / B, 4), round(tf * 1e3 / B, 4), kept, roof


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
    from efficientdet.pytorch_amd import EFFICIENTDET, ddp
    import torch.distributed as dist
    a.ddp_graph_ok, probe_note = False, None
    if world > 1 or a.ddp_single:
        backend = os.environ.get('EFFDET_BENCH_BACKEND', 'nccl')                             # 'nccl' IS RCCL on ROCm
        if backend == 'nccl' and not a.no_ddp_graph and not a.no_graph and not a.torch_optim:
            # pre-flight, BEFORE this job owns a communicator: can this box capture + replay an RCCL collective inside a hipGraph?
            # (N > 1: the ranks' probe children form their own N-rank group next to the job's port -- the real xGMI all-reduce is what
            #  gets captured, not a single-rank copy)
            if world > 1:
                ok, probe_note = ddp.rccl_graph_probe(local, rank=rank, world_size=world,
                                                      port=ddp.probe_port(os.environ.get('MASTER_PORT', '29500')))
            else:
                ok, probe_note = ddp.rccl_graph_probe(local)
        else:
            ok, probe_note = False, 'not probed (%s)' % ('backend %s' % backend if backend != 'nccl' else 'graph capture disabled by flag')
        if a.ddp_single:
            os.environ.setdefault('RANK', '0'); os.environ.setdefault('WORLD_SIZE', '1'); os.environ.setdefault('LOCAL_RANK', '0')
        ddp.init_process_group_from_env(backend)
        flag = torch.tensor([1 if ok else 0], device=dev)                                   # every rank takes the same path
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        a.ddp_graph_ok = bool(int(flag.item()))
    cfg = EFFICIENTDET[a.network]
    d0_512 = a.network == 'efficientdet-d0' and a.size == 512

    MODE_NOTE = {
        'f32_bf16x3': 'fp32 storage, bf16x3 MFMA products (hi*hi + hi*lo + lo*hi, fp32 accumulate): class probabilities within 1e-3 element-relative, '
                      'box deltas / taps 2.5e-3, losses / gradient norms within 1e-3 of the real reference (tests/test_gpu_model.py) -- the parity-qualified headline mode',
        'f32': 'fp32 storage, exact-fp32 MFMA products (v_mfma_f32_16x16x4_f32): the strict parity mode (1e-3 element-relative)',
        'bf16': 'bf16 storage + bf16 MFMA products: throughput mode, gated at 2.5e-2 of tensor scale (10 % D4) -- NOT a parity mode',
    }
    EXTRA_KEY = {'f32_bf16x3': 'parity_mode_bf16x3', 'f32': 'strict_mode_f32', 'bf16': 'throughput_mode_bf16'}
    tdt = {'bf16': torch.bfloat16, 'f32': torch.float32, 'f32_bf16x3': torch.float32}
    arith = {'bf16': 'f32', 'f32': 'f32', 'f32_bf16x3': 'bf16x3'}
    others = [m for m in ('f32_bf16x3', 'f32', 'bf16') if m != a.dtype]

    value, ms_step, final_loss, roof, img, graphed, host_ms = train_leg(a, a.dtype, a.steps, a.warmup, rank, world, local, dev, not a.no_roofline)
    out = {
        'metric': 'images/sec EfficientDet-D0 512px fwd+bwd', 'value': round(value, 2), 'unit': 'images/sec', 'n_gpus': world,
        'steps': a.steps, 'warmup': a.warmup, 'ms_per_step': round(ms_step, 3), 'higher_is_better': True, 'scaling': 'weak',
        'vs_baseline': None, 'dtype': a.dtype, 'data': 'synthetic',
        'config': {'workload': 'EfficientDet-D0 train step (fwd + FocalLoss/SmoothL1 + bwd + clip_grad_norm + AdamW), batch %d/GPU @ %dx%d, '
                               'synthetic COCO-shape targets, 80 classes, random-init, W_bifpn=%d D_bifpn=%d, drop_connect 0.2 active'
                               % (a.batch, a.size, a.size, cfg['W_bifpn'], cfg['D_bifpn']),
                   'network': a.network, 'global_batch': a.batch * world, 'image_size': a.size,
                   'parallelism': 'dp%d' % world + (' (world_size-1 RCCL group: ddp.wrap + bucketed all-reduce on the one GPU)' if a.ddp_single else ''),
                   'ddp_graph': ({'captured': bool(graphed), 'rccl_capture_probe': probe_note} if (world > 1 or a.ddp_single) else None),
                   'arithmetic': MODE_NOTE[a.dtype],
                   'final_loss': round(final_loss, 4), 'launch': 'hipGraph replay (one graph launch per step)' if graphed else 'eager launches'},
        'algorithmic_tflops_per_gpu': round(TRAIN_GFLOP_PER_IMG * a.batch / ms_step, 2) if d0_512 else None,
        # host time to ISSUE one step, per rank (wall time per step is ms_per_step): host ~= wall means the launch path, not the GPU, is the bound
        'host_ms_per_step': host_ms,
    }
    if roof is not None:
        out['roofline'] = roof

    if world == 1 and not a.no_extra_modes:
        # the SAME workload in the other two arithmetic modes, >= 20 timed steps each, with their own rooflines
        # (exact fp32 against the 157.3 TFLOP/s fp32 MFMA peak, bf16 against 2500, bf16x3 against 2500 / 3 algorithmic)
        for mode in others:
            pv, pms, ploss, proof, _, pgr, _ = train_leg(a, mode, a.extra_steps, a.extra_warmup, rank, world, local, dev, not a.no_roofline)
            out[EXTRA_KEY[mode]] = {'dtype': mode, 'value': round(pv, 2), 'unit': 'images/sec', 'ms_per_step': round(pms, 3),
                                    'steps': a.extra_steps, 'warmup': a.extra_warmup, 'final_loss': round(ploss, 4),
                                    'algorithmic_tflops_per_gpu': round(TRAIN_GFLOP_PER_IMG * a.batch / pms, 2) if d0_512 else None,
                                    'launch': 'hipGraph replay' if pgr else 'eager launches', 'note': MODE_NOTE[mode], 'roofline': proof}

    if rank == 0 and world == 1 and not a.no_inference:
        ti, tf, kept, iroof = inference_leg(a.network, tdt[a.dtype], dev, img, reps=a.infer_reps, graph=not a.no_graph, f32_arith=arith[a.dtype],
                                            dtype_name=None if a.no_roofline else a.dtype)
        gf = INFER_GFLOP_PER_IMG.get((a.network, a.size))
        out['inference'] = {'workload': 'configs[1]: D0 eval batch %d @ %d: forward + decode + per-image NMS (thr 0.01, IoU 0.5)' % (a.batch, a.size),
                            'dtype': a.dtype, 'ms_per_img': ti, 'forward_only_ms_per_img': tf, 'kept_boxes_img0': kept, 'reps': a.infer_reps,
                            'forward_tflops': round(gf / tf, 2) if gf else None, 'roofline': iroof,
                            'launch': 'eager launches' if a.no_graph else 'forward + decode + NMS + gather as ONE hipGraph replay (end-to-end number; forward_only is eager)'}
        if not a.no_extra_modes:
            for mode in others:
                ti, tf, kept, _ = inference_leg(a.network, tdt[mode], dev, img, reps=a.infer_reps, graph=not a.no_graph, f32_arith=arith[mode])
                out['inference'][EXTRA_KEY[mode]] = {'dtype': mode, 'ms_per_img': ti, 'forward_only_ms_per_img': tf, 'kept_boxes_img0': kept,
                                                     'reps': a.infer_reps}
        del img
        torch.cuda.empty_cache()
        if not a.no_d4:
            from efficientdet.pytorch_amd.synthetic import synthetic_batch
            img4 = synthetic_batch(8, 1024, seed=1, num_classes=80)[0].to(dev)
            ti, tf, kept, iroof = inference_leg('efficientdet-d4', tdt[a.dtype], dev, img4, reps=a.infer_reps, graph=not a.no_graph,
                                                f32_arith=arith[a.dtype], dtype_name=None if a.no_roofline else a.dtype)
            out['inference_d4'] = {'workload': 'configs[4]: D4 eval batch 8 @ 1024: forward + decode + per-image NMS (thr 0.01, IoU 0.5)',
                                   'dtype': a.dtype, 'ms_per_img': ti, 'forward_only_ms_per_img': tf, 'kept_boxes_img0': kept, 'reps': a.infer_reps,
                                   'forward_tflops': round(455.596 / tf, 2), 'roofline': iroof}
            if not a.no_extra_modes:
                for mode in others:
                    ti, tf, kept, _ = inference_leg('efficientdet-d4', tdt[mode], dev, img4, reps=max(a.infer_reps // 2, 2), graph=not a.no_graph,
                                                    f32_arith=arith[mode])
                    out['inference_d4'][EXTRA_KEY[mode]] = {'dtype': mode, 'ms_per_img': ti, 'forward_only_ms_per_img': tf, 'kept_boxes_img0': kept,
                                                            'reps': max(a.infer_reps // 2, 2)}

    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        out['cpu_baseline'] = cpu_baseline(a.network, a.size)

    if rank == 0:
        print(json.dumps(out))
    if world > 1 or a.ddp_single:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
