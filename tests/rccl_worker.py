"""Worker of tests/test_gpu_rccl.py: the PRODUCT data-parallel path on RCCL itself (backend 'nccl' IS RCCL on ROCm), as a
world_size = 1 process group on the one GPU a test box has.  One rank is enough to execute what no gloo test reaches: RCCL
communicator bring-up on the device, ddp.wrap's bucket hooks firing RCCL all-reduces on RCCL's stream from the custom autograd
nodes, ClipAdamW over the bucket views, and the whole DDP step -- collectives included -- captured as ONE hipGraph and replayed
(graph.GraphedTrainStep after DDP's bucket rebuild).  The path this replaces: train.py:237-258 of the reference.

Stages write their results into the JSON as they finish, so a crash in a later stage still shows how far RCCL got."""
import json
import os
import sys

os.environ.setdefault('TORCH_NCCL_ASYNC_ERROR_HANDLING', '0')      # (capture rule of torch DDP + graphs; harmless on current builds)
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    out_path, arith = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else 'f32_hf16x3_bwd_bf16x3')
    from efficientdet.pytorch_amd import EfficientDet, EFFICIENTDET, ddp, synthetic_batch
    from efficientdet.pytorch_amd.graph import GraphedTrainStep
    from efficientdet.pytorch_amd.optim import ClipAdamW
    res = {'stage': 'init', 'arith': arith}

    def dump():
        json.dump(res, open(out_path, 'w'))
    dump()
    torch.cuda.set_device(0)
    ddp.init_process_group_from_env('nccl', for_capture=True)      # stage 3 captures the DDP step
    assert dist.get_backend() == 'nccl' and dist.get_world_size() == 1
    res['rccl_version'] = list(torch.cuda.nccl.version())
    # ---- stage 1: a bare RCCL collective on the device
    t = torch.arange(1024, device='cuda', dtype=torch.float32)
    dist.all_reduce(t); dist.barrier(); torch.cuda.synchronize()
    res['allreduce_ok'] = bool(torch.equal(t.cpu(), torch.arange(1024, dtype=torch.float32)))
    res['stage'] = 'collective'; dump()

    net, nc, lr = 'efficientdet-d0', 20, 1e-4
    c = EFFICIENTDET[net]
    img, ann = synthetic_batch(4, 128, seed=3, num_classes=nc)
    img, ann = img.cuda(), ann.cuda()

    def build(wrap):
        torch.manual_seed(0)
        m = EfficientDet(nc, network=net, W_bifpn=c['W_bifpn'], D_bifpn=c['D_bifpn'], D_class=c['D_class'], compute_dtype=torch.float32,
                         f32_arith=arith).cuda()
        m._dc.update(seed=1234, step=0)                      # drop_connect ACTIVE, same Philox stream in every model of this test
        m.train(); m.is_training = True; m.freeze_bn()
        ddp.freeze_dead_parameters(m)
        if wrap == 'capture':
            w = ddp.wrap_for_capture(m, device_ids=[0])      # built on the stream GraphedTrainStep will warm up and capture on
        elif wrap:
            w = ddp.wrap(m, device_ids=[0])
        else:
            w = m
        opt = ClipAdamW([p for p in m.parameters() if p.requires_grad], lr=lr, max_norm=0.1)
        return m, w, opt

    def step(w, opt):
        opt.zero_grad(set_to_none=True)
        cl, rl = w([img, ann])
        (cl.mean() + rl.mean()).backward()
        opt.step()
        return float(cl) + float(rl)

    def flat(m):
        return torch.cat([p.detach().reshape(-1) for p in m.parameters()]).clone()

    # ---- stage 2: eager steps under ddp.wrap over RCCL == the bare module, bit for bit (the all-reduce average over one rank is
    #      the identity; bucket views, hooks and the RCCL stream hand-over must not change a single bit)
    mp, wp, op_ = build(False)
    md, wd, od = build(True)
    lp = [step(wp, op_) for _ in range(3)]
    ld = [step(wd, od) for _ in range(3)]
    torch.cuda.synchronize()
    res.update(eager_losses_plain=lp, eager_losses_ddp=ld, eager_bitwise=bool(torch.equal(flat(mp), flat(md))),
               finite=bool(torch.isfinite(flat(md)).all()), bucket_views=bool(od._table['g_last'] is not None),
               prep_replay=bool(next(iter(md._prep.values())).replay))
    res['stage'] = 'eager'; dump()
    del mp, wp, op_

    # ---- stage 3: the DDP step captured as a hipGraph with its RCCL all-reduces inside (11 eager iterations first: DDP rebuilds
    #      its buckets after the first one and torch asks for 11 before a capture), replays vs the same number of eager DDP steps
    me, we, oe = build(True)
    mg, wg, og = build('capture')
    WARM, REPLAYS = 11, 3
    le = [step(we, oe) for _ in range(WARM + REPLAYS)]
    torch.cuda.synchronize()
    res['stage'] = 'capture'; dump()
    g = GraphedTrainStep(wg, og, img, ann, warmup=WARM)
    res['stage'] = 'captured'; dump()
    lg = []
    for _ in range(REPLAYS):
        cl, rl = g(); lg.append(float(cl) + float(rl))
    torch.cuda.synchronize()
    d = (flat(me) - flat(mg)).abs()
    res.update(graph_losses=lg, eager_tail_losses=le[WARM:], graph_param_diff_mean=float(d.mean()), graph_param_diff_max=float(d.max()),
               graph_finite=bool(torch.isfinite(flat(mg)).all()), lr=lr, steps=WARM + REPLAYS)
    # ONE replay vs ONE eager DDP step from the same restored state (graph.replay_vs_eager): the strict form of 'replays track eager'
    from efficientdet.pytorch_amd.graph import replay_vs_eager
    res['replay_vs_eager'] = replay_vs_eager(g)
    res['stage'] = 'checked'; dump()
    # a second batch through the static input buffers, one more replay: still a real step
    g.images.copy_(img.flip(0)); g.annotations.copy_(ann.flip(0))
    cl, rl = g(); torch.cuda.synchronize()
    res['graph_second_batch_loss'] = float(cl) + float(rl)
    res['stage'] = 'done'; dump()
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
