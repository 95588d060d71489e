"""Whole-step hipGraph capture of the reference's training iteration (train.py:95-118: zero_grad -> forward -> loss ->
backward -> clip -> optimizer.step) for fixed batch shapes.

The eager step issues ~560 launches through Python + ctypes (host floor ~12 ms on D0 B=32); a captured step is ONE
hipGraphLaunch.  Everything the step does is already capture-safe by construction: kernels never allocate or synchronise
(include/effdet_hip.h), the per-step parameter repacks replay a recorded table, the drop_connect step counter and the AdamW
step counters live on the device, and the gradient-pointer table is static inside the graph's private memory pool.  The optimizer's hyper-parameters (lr, betas,
eps, weight_decay, max_norm) are read by the update kernel from a device buffer that __call__ refreshes from
optimizer.param_groups before every replay, so lr schedulers keep working; only switching clipping on/off needs a re-capture.

    step = GraphedTrainStep(model, optimizer, images, annotations)       # warm-up + capture
    for batch in loader:
        step.images.copy_(batch_images); step.annotations.copy_(batch_annots)
        cls_loss, reg_loss = step()                                       # device tensors, valid until the next call

Under DistributedDataParallel (one process per GPU) build the module with ddp.wrap_for_capture and give the constructor
warmup >= 11: DDP's bucketed RCCL all-reduces are then captured with the step (tests/test_gpu_rccl.py).  Drop every reference to losses of earlier EAGER steps
before constructing this (a live loss keeps that step's autograd graph and its default-stream AccumulateGrad nodes alive, which
a capture on another stream must not touch)."""
import torch

from . import ops


def _build_pending_tables(model):
    """The first step of a model RECORDS its parameter-preparation jobs; the device-side job table is built at the start of the
    next step -- which must not be the captured one (building allocates and uploads).  Build it now."""
    while isinstance(model, (torch.nn.DataParallel, torch.nn.parallel.DistributedDataParallel)):
        model = model.module
    for prep in getattr(model, '_prep', {}).values():
        if prep.dirty:
            prep._build()


def _drain_collective_watchdog(seconds=None):
    """ProcessGroupNCCL's watchdog thread retires finished collectives by polling their end events every ~100 ms.  The warm-up's
    all-reduces are finished here (device synchronised) but not necessarily RETIRED -- and once the capture below pulls RCCL's stream
    into capture mode, HIP refuses hipEventQuery on an event whose stream is capturing (`hipErrorCapturedEvent`, "operation not
    permitted on an event last recorded in a capturing stream") even though the record itself happened before the capture: the
    watchdog throws and the process aborts.  Seen in about one bench run out of four (round 4).  CUDA answers such a query, which is why
    the stock eager-warm-up -> capture recipe carries no wait; here the watchdog gets ten of its polling periods to empty its list."""
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        import os
        import time
        time.sleep(float(os.environ.get('EFFDET_WATCHDOG_DRAIN_S', '1.0')) if seconds is None else seconds)     # (0: the A/B that shows the race)
        torch.cuda.synchronize()


class GraphedTrainStep:
    def __init__(self, model, optimizer, images, annotations, warmup=2, clip_fn=None):
        if not images.is_cuda:
            raise RuntimeError('GraphedTrainStep needs GPU-resident batches')
        self.model, self.optimizer = model, optimizer
        self.images, self.annotations = images.clone(), annotations.clone()      # static input buffers of the graph
        self.clip_fn = clip_fn                  # e.g. lambda: clip_grad_norm_(params, 0.1) for stock optimizers (ClipAdamW clips itself)
        # ONE side stream for the warm-up AND the capture; a DDP module built by ddp.wrap_for_capture brings the stream its
        # AccumulateGrad nodes / bucket hooks already live on
        side = getattr(model, '_effdet_capture_stream', None) or torch.cuda.Stream()
        self.stream = side                      # eager steps of this model belong on it too (its AccumulateGrad nodes live there)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):           # warm-up on the side stream: records the ParamPrep table, sizes the zero pool, and
            for _ in range(max(1, warmup)):     # (re)creates the AccumulateGrad nodes off the legacy default stream, which a
                self._step()                    # capture must not touch
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        _drain_collective_watchdog()
        _build_pending_tables(model)            # (a table recorded by the warm-up is built here: no allocation / H2D copy under capture)
        self.graph = torch.cuda.CUDAGraph()
        # Under torch.distributed the process group's watchdog THREAD polls the events of earlier collectives (hipEventQuery); in
        # the default 'global' capture mode that call from another thread is "not permitted when stream is capturing" and kills
        # the process (seen with a bare all-reduce capture right behind eager collectives).  'thread_local' confines the
        # unsafe-call check to the capturing thread.
        dist_on = torch.distributed.is_available() and torch.distributed.is_initialized()
        with torch.cuda.graph(self.graph, stream=side, capture_error_mode='thread_local' if dist_on else 'global'):
            self.losses = self._step()
        torch.cuda.synchronize()
        self._hyper0 = self._hyper_sig()
        ops.bump_param_generation()

    def _step(self):
        self.optimizer.zero_grad(set_to_none=True)
        cl, rl = self.model([self.images, self.annotations])
        (cl.mean() + rl.mean()).backward()
        if self.clip_fn is not None:
            self.clip_fn()
        self.optimizer.step()
        return cl, rl

    def __call__(self):
        # hyper-parameters (lr schedule, train.py:133,269) live in a device buffer the captured update kernel reads: refresh
        # it from param_groups before the replay; stock optimizers (by-value kernel scalars) cannot follow -> fail loudly
        sync = getattr(self.optimizer, 'sync_hyper', None)
        if sync is not None:
            sync()
        elif self._hyper_sig() != self._hyper0:
            raise RuntimeError('GraphedTrainStep: optimizer hyper-parameters changed after capture and %s bakes them into the '
                               'captured kernels; use optim.ClipAdamW or re-capture the step' % type(self.optimizer).__name__)
        self.graph.replay()
        ops.bump_param_generation()             # the replay rewrote the parameters through raw pointers (no Tensor._version bump)
        # the replay's memcpy node re-installed the GRAPH's gradient-pointer table on the device: an eager ClipAdamW.step() that follows
        # must upload its own pointers again even when p.grad sits at the addresses of the last eager step (the usual case after
        # zero_grad(set_to_none=True)) -- otherwise its kernels would read the graph pool's gradients
        tbl = getattr(self.optimizer, '_table', None)
        if tbl is not None:
            tbl['g_last'] = None
        return self.losses

    def _hyper_sig(self):
        return [sorted((k, repr(v)) for k, v in g.items() if k != 'params') for g in self.optimizer.param_groups]


def replay_vs_eager(graphed, eager_step=None):
    """Is a replay of `graphed` (a GraphedTrainStep over optim.ClipAdamW) the eager step?  From ONE saved state -- parameters, Adam
    moments + per-tensor step counters, the drop_connect counter, all restored IN PLACE so the graph's pointers stay valid -- run the eager
    step (default: graphed._step on the capture stream) and the replay, each twice, on the graph's static batch, and compare the
    parameters they produce.  The path has no float atomics, so eager == eager and replay == replay bit for bit (which also proves the
    restore complete), and replay vs eager differs at most by what a collective's summation order may change under DDP.
    -> {'replay_vs_eager', 'eager_vs_eager', 'replay_vs_replay'} (norm of the parameter difference / norm of the step's own update),
       'update_norm', 'finite', 'losses_replay', 'losses_eager'.   Leaves the model one replayed step past the state it found.
    (Round 5: this is the check that found the captured step training on ~1e-7 of its gradients -- a hipMemsetAsync node inside the loss
    did not hold in the graph -- while every 'losses track, parameters within AdamW's sign noise' test stayed green.)"""
    opt = graphed.optimizer
    model = graphed.model
    while isinstance(model, (torch.nn.DataParallel, torch.nn.parallel.DistributedDataParallel)):
        model = model.module
    if not hasattr(opt, 'exp_avg') or getattr(opt, '_table', None) is None:
        raise RuntimeError('replay_vs_eager needs optim.ClipAdamW (its state lives in arenas that can be restored in place)')
    ps = [p for p in opt._table['params']]
    t = opt._table
    dc = [v for k, v in getattr(model, '_dc', {}).items() if isinstance(k, tuple) and k[0] == 'step_dev']

    def flat():
        return torch.cat([p.detach().reshape(-1).float() for p in ps]).clone()
    torch.cuda.synchronize()
    p0, m0, v0, s0 = flat(), opt.exp_avg.clone(), opt.exp_avg_sq.clone(), t['steps'].clone()
    dc0 = [d.clone() for d in dc]
    side = graphed.stream

    def run(replay):
        with torch.no_grad():
            o = 0
            for p in ps:
                n = p.numel(); p.copy_(p0[o:o + n].view_as(p)); o += n
            opt.exp_avg.copy_(m0); opt.exp_avg_sq.copy_(v0); t['steps'].copy_(s0)
            for d, d0 in zip(dc, dc0):
                d.copy_(d0)
        torch.cuda.synchronize()
        if replay:
            cl, rl = graphed()
        else:
            with torch.cuda.stream(side):
                cl, rl = (graphed._step if eager_step is None else eager_step)()
        torch.cuda.synchronize()
        return flat(), (float(cl.detach().reshape(-1)[0]), float(rl.detach().reshape(-1)[0]))
    pe, le = run(False); pe2, _ = run(False)
    pg, lg = run(True); pg2, _ = run(True)
    upd = float((pe - p0).norm())
    rel = lambda a, b: float((a - b).norm()) / max(upd, 1e-30)
    return {'replay_vs_eager': rel(pg, pe), 'eager_vs_eager': rel(pe2, pe), 'replay_vs_replay': rel(pg2, pg), 'update_norm': upd,
            'finite': bool(torch.isfinite(pg).all()), 'losses_replay': lg, 'losses_eager': le}


class GraphedDetect:
    """Eval forward + decode + on-device NMS + gather (models/efficientdet.py:57-86 for every image of the batch) captured as ONE
    hipGraph for a fixed batch shape: the ~270 launches of the network and the ~90 of the post-processing become one graph launch.
    (Round 4: the NMS's sort is an in-tree radix sort whose launch geometry depends on shapes only; the rocPRIM sort it replaces
    faulted on the second replay of a captured call, which had kept sort + NMS + gather outside the graph.)

        det = GraphedDetect(model, images)            # warm-up + capture
        det.images.copy_(batch); results = det()      # -> [(scores[K], labels[K] int64, boxes[K,4]) per image], score-descending;
                                                      #    fresh tensors every call (they survive the next replay)
    """

    def __init__(self, model, images, warmup=2):
        from . import ops
        if not images.is_cuda:
            raise RuntimeError('GraphedDetect needs GPU-resident batches')
        self.model, self.ops = model, ops
        self.images = images.clone()
        H, W = int(images.shape[2]), int(images.shape[3])

        def run():
            with torch.no_grad():
                cls, reg, anc = model.forward_raw(self.images)
                boxes, score, label = ops.decode_score(anc, reg, cls, H, W)
                idx, count = ops.nms(boxes, score, float(model.threshold), float(model.iou_threshold))
                s, l, b = ops.gather_dets(boxes, score, label, idx, count)
                return s, l, b, count
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(max(1, warmup)):
                run()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        _build_pending_tables(model)
        self.graph = torch.cuda.CUDAGraph()
        self.thresholds = (float(model.threshold), float(model.iou_threshold))       # baked into the captured launches
        # (inside a torch.distributed job the process group's watchdog thread polls events while this capture is open: 'thread_local'
        #  confines the unsafe-call check to this thread, as in GraphedTrainStep)
        dist_on = torch.distributed.is_available() and torch.distributed.is_initialized()
        with torch.cuda.graph(self.graph, capture_error_mode='thread_local' if dist_on else 'global'):
            self.s, self.l, self.b, self.count = run()
        torch.cuda.synchronize()

    def __call__(self):
        m = self.model
        if (float(m.threshold), float(m.iou_threshold)) != self.thresholds:
            raise RuntimeError('GraphedDetect: model.threshold / iou_threshold changed after capture (they are kernel arguments of the '
                               'captured NMS); build a new GraphedDetect')
        self.graph.replay()
        counts = self.count.tolist()                       # the one device->host sync (the reference syncs too)
        if ops.MODEL_ARITH[getattr(m, 'f32_arith', 'f32')][2] == 'f16x3':
            ops.check_range_flag(self.s.device)            # f16x3: an activation beyond fp16's range is an error, not a plausible score
        # the graph's output buffers are rewritten by the next replay: hand out COPIES of the kept rows (one clone of the rows up to
        # the largest count), so an evaluator that accumulates results over batches keeps what it was given -- like model.detect
        k = max(counts) if counts else 0
        s, l, b = self.s[:, :k].clone(), self.l[:, :k].clone(), self.b[:, :k].clone()
        return [(s[i, :n], l[i, :n], b[i, :n]) for i, n in enumerate(counts)]
