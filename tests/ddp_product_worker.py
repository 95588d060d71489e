"""Worker of tests/test_gpu_ddp_product.py: one rank of a world-size-2 data-parallel run of the HIP model (the PRODUCT path:
ddp.wrap + ParamPrep replay + custom autograd nodes under DDP's bucket hooks + ClipAdamW over bucket views).  Both ranks
share the one visible GPU, so the collective backend is gloo (RCCL needs one device per rank); the control flow, hooks and
bucket views are exactly those of the N-GPU run (train.py:237-258 of the reference is the path this replaces)."""
import json
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    out_path, dtype_name = sys.argv[1], sys.argv[2]
    wrap = sys.argv[3] if len(sys.argv) > 3 else 'ddp.wrap'
    from efficientdet.pytorch_amd import EfficientDet, EFFICIENTDET, ddp, synthetic_batch
    from efficientdet.pytorch_amd.optim import ClipAdamW
    rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
    torch.cuda.set_device(0)
    ddp.init_process_group_from_env('gloo')
    dtype = torch.float32 if dtype_name == 'f32' else torch.bfloat16
    net, nc = 'efficientdet-d0', 20
    c = EFFICIENTDET[net]

    def build():
        torch.manual_seed(0)
        m = EfficientDet(nc, network=net, W_bifpn=c['W_bifpn'], D_bifpn=c['D_bifpn'], D_class=c['D_class'], compute_dtype=dtype).cuda()
        m.backbone.drop_connect_rate = 0.0
        m.train(); m.is_training = True; m.freeze_bn()
        return m
    model = build()
    if wrap == 'reference':        # train.py:250-251 as written: every parameter trainable, unused-parameter search on
        net_ddp = torch.nn.parallel.DistributedDataParallel(model, device_ids=[0], find_unused_parameters=True)
    else:
        net_ddp = ddp.wrap(model, device_ids=[0])
    params = [p for p in model.parameters() if p.requires_grad]
    opt = ClipAdamW(params, lr=1e-4, max_norm=0.1)
    img, ann = synthetic_batch(2 * world, 128, seed=3, num_classes=nc)
    simg, sann = ddp.shard_batch(img, ann, rank, world)
    simg, sann = simg.cuda(), sann.cuda()
    res = {'rank': rank, 'world': world, 'dtype': dtype_name, 'wrap': wrap}
    # ---- step 1: DDP-averaged shard gradients == single-process full-batch gradients
    opt.zero_grad(set_to_none=True)
    cl, rl = net_ddp([simg, sann])
    (cl.mean() + rl.mean()).backward()
    torch.cuda.synchronize()
    twin = build()
    tcl, trl = twin([img.cuda(), ann.cuda()])
    (tcl.mean() + trl.mean()).backward()
    torch.cuda.synchronize()
    worst, worst_l2, nchk = 0.0, 0.0, 0
    for (k, p), (_, q) in zip(model.named_parameters(), twin.named_parameters()):
        if q.grad is None:
            assert p.grad is None or not p.requires_grad or not bool(p.grad.any()), k
            continue
        scale = float(q.grad.abs().max()) + 1e-12
        worst = max(worst, float((p.grad - q.grad).abs().max()) / scale); nchk += 1
        worst_l2 = max(worst_l2, float((p.grad.double() - q.grad.double()).norm()) / (float(q.grad.double().norm()) + 1e-12))
    res.update(grad_worst_rel=worst, grad_worst_l2=worst_l2, grad_tensors=nchk, loss=float(cl) + float(rl))
    # ---- 3 optimizer steps over bucket views, never synchronising in between; replicas must stay bit-identical
    opt.step()
    for it in range(2):
        opt.zero_grad(set_to_none=(wrap == 'reference'))          # ddp.wrap: bucket views stay in place (gradient_as_bucket_view)
        cl, rl = net_ddp([simg, sann])
        (cl.mean() + rl.mean()).backward()
        opt.step()
    torch.cuda.synchronize()
    flat = torch.cat([p.detach().reshape(-1) for p in params])
    mine = flat.double().sum().reshape(1).cpu()
    both = [torch.zeros(1, dtype=torch.float64) for _ in range(world)]
    dist.all_gather(both, mine)
    res.update(param_checksums=[float(b) for b in both], finite=bool(torch.isfinite(flat).all()), final_loss=float(cl) + float(rl),
               prep_replay=bool(next(iter(model._prep.values())).replay), ptr_uploads_skipped=opt._table['g_last'] is not None)
    dist.barrier()
    if rank == 0:
        json.dump(res, open(out_path, 'w'))
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
