"""CPU tier: the N>1 data-parallel path (efficientdet.pytorch_amd.ddp) with world_size 2 over gloo.

The product model has no CPU path, so the compute under the DDP wrapper is the ORACLE driven by the product module's
own parameter containers (same state_dict layout): what is tested is the wrapper logic -- dead-parameter freezing (no
find_unused_parameters), equal sharding, bucketed all-reduce averaging == single-process full-batch gradients."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn

from oracle import effdet_oracle as O

NET, NC, S = 'efficientdet-d0', 4, 128


class OracleDriven(nn.Module):
    def __init__(self):
        super().__init__()
        from efficientdet.pytorch_amd import EfficientDet, EFFICIENTDET
        c = EFFICIENTDET[NET]
        self.model = EfficientDet(NC, network=NET, W_bifpn=c['W_bifpn'], D_bifpn=c['D_bifpn'])
        self.model.load_state_dict(O.make_state_dict(NET, NC, seed=0))

    def forward(self, inputs):
        img, ann = inputs
        sd = dict(self.model.named_buffers()); sd.update(dict(self.model.named_parameters()))
        return O.train_losses(sd, NET, NC, img, ann)


def _free_port():
    s = socket.socket(); s.bind(('127.0.0.1', 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(2)
    from efficientdet.pytorch_amd import ddp
    ddp.init_process_group_from_env('gloo')
    m = OracleDriven()
    nfrozen = ddp.freeze_dead_parameters(m.model)
    net = ddp.wrap(m, device_ids=None)
    img, ann = O.synthetic_batch(2 * world, S, seed=1, num_classes=NC)
    si, sa = ddp.shard_batch(img, ann, rank, world)
    cl, rl = net([si, sa])
    (cl.mean() + rl.mean()).backward()
    if rank == 0:
        g = {k: p.grad.clone() for k, p in m.model.named_parameters() if p.grad is not None}
        torch.save({'grads': g, 'nfrozen': nfrozen, 'shard': tuple(si.shape)}, out)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_ddp_world2_gloo_matches_full_batch(tmp_path):
    world = 2
    out = str(tmp_path / 'r0.pt')
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    res = torch.load(out)
    assert res['nfrozen'] == 5 and res['shard'] == (2, 3, S, S)
    # single-process reference on the FULL batch (mean over images == mean of the two equal shards' means)
    m = OracleDriven()
    img, ann = O.synthetic_batch(2 * world, S, seed=1, num_classes=NC)
    cl, rl = m([img, ann])
    (cl.mean() + rl.mean()).backward()
    from efficientdet.pytorch_amd import ddp
    dead = set(ddp.dead_parameter_names())
    n = 0
    for k, p in m.model.named_parameters():
        if k in dead:
            assert k not in res['grads']
            continue
        ref, got = p.grad, res['grads'][k]
        assert float((got - ref).abs().max()) <= 2e-4 * float(ref.abs().max()) + 1e-9, k
        n += 1
    assert n == len(m.model.live_parameters()) == 274


def test_shard_batch_is_an_equal_partition():
    from efficientdet.pytorch_amd import ddp
    img = torch.arange(8 * 3).float().view(8, 3, 1, 1); ann = torch.arange(8).float().view(8, 1, 1)
    parts = [ddp.shard_batch(img, ann, r, 4) for r in range(4)]
    assert torch.equal(torch.cat([p[0] for p in parts]), img) and all(p[0].shape[0] == 2 for p in parts)
