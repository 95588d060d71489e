"""Data-parallel training over RCCL / xGMI: one process per GPU, DistributedDataParallel all-reduce
overlapped with backward (the path of the reference's train.py:237-258, minus its defects).

The model is a full replica per GPU; BatchNorm is frozen (no SyncBN traffic); the per-rank loss is a local
batch mean, so DDP's gradient average is the global mean for equal shards.  The one exchange step is the
bucketed all-reduce of 39.3 MB of live fp32 gradients (D0), launched from autograd hooks as each of the
~20 autograd nodes (head+loss, neck, 16 MBConv blocks, stem) finishes, i.e. the head's 22 MB are on the
wire while the neck/backbone backward still runs.  The five never-executed backbone tensors
(_conv_head, _bn1, _fc) are frozen so that find_unused_parameters (a graph walk per step in the
reference, train.py:251) is not needed.
"""
import os

import torch
import torch.distributed as dist


def init_process_group_from_env(backend=None):
    """env:// rendezvous (RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT), NCCL(=RCCL) on GPU, gloo on CPU."""
    if dist.is_initialized():
        return
    if backend is None:
        backend = 'nccl' if torch.cuda.is_available() else 'gloo'
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29500')
    dist.init_process_group(backend=backend, init_method='env://')


def dead_parameter_names():
    return ('backbone._conv_head.weight', 'backbone._bn1.weight', 'backbone._bn1.bias', 'backbone._fc.weight',
            'backbone._fc.bias')


def freeze_dead_parameters(model):
    """requires_grad=False on the tensors the forward never touches (kept in state_dict for checkpoint ABI)."""
    names = set(dead_parameter_names())
    n = 0
    for k, p in model.named_parameters():
        if k in names:
            p.requires_grad_(False); n += 1
    return n


def wrap(model, device_ids=None, bucket_cap_mb=8):
    """DistributedDataParallel with small buckets (the head's gradients are ready first and should leave early),
    bucket views instead of copies, and no unused-parameter search."""
    freeze_dead_parameters(model)
    return torch.nn.parallel.DistributedDataParallel(model, device_ids=device_ids, bucket_cap_mb=bucket_cap_mb,
                                                     gradient_as_bucket_view=True, find_unused_parameters=False,
                                                     broadcast_buffers=False)


def shard_batch(images, annotations, rank, world_size):
    """Contiguous equal shards of a global batch (the reference divides batch_size by ngpus, train.py:247)."""
    per = images.shape[0] // world_size
    sl = slice(rank * per, (rank + 1) * per)
    return images[sl], annotations[sl]
