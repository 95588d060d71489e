"""Round-5 probe (world_size-1 RCCL group on one GPU): does a replay of the captured DDP step do what an eager step does?
bench.py's self-check said no (the replay's update looked like an update from ZERO gradients); this separates the suspects:
  A  captured DDP step, no eager step after the capture: replay vs a plain twin model's eager step from the same state
  B  the same model: eager DDP step from the restored state vs the twin (must be equal), then another replay (did the eager step in
     between change what the replay does?)
  C  plain (no DDP) captured step: replay vs eager from the same state
Run:  RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29517 python tools/ddp_graph_probe.py"""
import os
import sys

os.environ.setdefault('RANK', '0'); os.environ.setdefault('WORLD_SIZE', '1'); os.environ.setdefault('LOCAL_RANK', '0')
os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29517')
os.environ.setdefault('TORCH_NCCL_ASYNC_ERROR_HANDLING', '0')
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from efficientdet.pytorch_amd import EfficientDet, EFFICIENTDET, ddp, synthetic_batch     # noqa: E402
from efficientdet.pytorch_amd.graph import GraphedTrainStep                                # noqa: E402
from efficientdet.pytorch_amd.optim import ClipAdamW                                       # noqa: E402

torch.cuda.set_device(0)
ddp.init_process_group_from_env('nccl', for_capture=True)
net, nc, lr = 'efficientdet-d0', 20, 1e-4
c = EFFICIENTDET[net]
img, ann = synthetic_batch(4, 128, seed=3, num_classes=nc)
img, ann = img.cuda(), ann.cuda()
ARITH = sys.argv[1] if len(sys.argv) > 1 else 'f32_bwd_bf16x3'


def build(wrap):
    torch.manual_seed(0)
    m = EfficientDet(nc, network=net, W_bifpn=c['W_bifpn'], D_bifpn=c['D_bifpn'], D_class=c['D_class'], compute_dtype=torch.float32,
                     f32_arith=ARITH).cuda()
    m._dc.update(seed=1234, step=0)
    m.train(); m.is_training = True; m.freeze_bn()
    ddp.freeze_dead_parameters(m)
    w = ddp.wrap_for_capture(m, device_ids=[0]) if wrap == 'capture' else (ddp.wrap(m, device_ids=[0]) if wrap else m)
    opt = ClipAdamW([p for p in m.parameters() if p.requires_grad], lr=lr, max_norm=0.1)
    return m, w, opt


def step(w, opt):
    opt.zero_grad(set_to_none=True)
    cl, rl = w([img, ann])
    (cl.mean() + rl.mean()).backward()
    opt.step()


def ps(m):
    return [p for p in m.parameters() if p.requires_grad]


def flat(m, grad=False):
    return torch.cat([(p.grad if grad else p).detach().reshape(-1).float() for p in ps(m)]).clone()


def snap(m, opt):
    torch.cuda.synchronize()
    dc = [v for k, v in m._dc.items() if isinstance(k, tuple) and k[0] == 'step_dev']
    return dict(p=flat(m), m=opt.exp_avg.clone(), v=opt.exp_avg_sq.clone(), s=opt._table['steps'].clone(), dc=[d.clone() for d in dc])


def load(m, opt, st):
    dc = [v for k, v in m._dc.items() if isinstance(k, tuple) and k[0] == 'step_dev']
    with torch.no_grad():
        o = 0
        for p in ps(m):
            n = p.numel(); p.copy_(st['p'][o:o + n].view_as(p)); o += n
        opt.exp_avg.copy_(st['m']); opt.exp_avg_sq.copy_(st['v']); opt._table['steps'].copy_(st['s'])
        for d, d0 in zip(dc, st['dc']):
            d.copy_(d0)
    torch.cuda.synchronize()


def rel(a, b, upd):
    return float((a - b).norm()) / max(upd, 1e-30)


# twin: plain eager model brought to the same state by the same number of steps is NOT needed -- it takes the state by copy
mt, wt, ot = build(False)
for _ in range(2):
    step(wt, ot)                                   # (builds the optimizer tables and the drop_connect counter)
torch.cuda.synchronize()

print('=== A: captured DDP step, nothing eager after the capture', flush=True)
mg, wg, og = build('capture')
g = GraphedTrainStep(wg, og, img, ann, warmup=11)
g(); torch.cuda.synchronize()
S0 = snap(mg, og)
g(); torch.cuda.synchronize()
Pg1, Gg1 = flat(mg), flat(mg, grad=True)
load(mt, ot, S0); step(wt, ot); torch.cuda.synchronize()
Pt, Gt = flat(mt), flat(mt, grad=True)
upd = float((Pt - S0['p']).norm())
print('update norm %.4e | replay vs twin eager: params %.3e of update | grads: |g_replay| %.4e |g_twin| %.4e |diff| %.4e'
      % (upd, rel(Pg1, Pt, upd), float(Gg1.norm()), float(Gt.norm()), float((Gg1 - Gt).norm())), flush=True)
tab = og._table
cur = torch.tensor([p.grad.data_ptr() if p.grad is not None else 0 for p in tab['params']], dtype=torch.int64)
print('optimizer pointer table == current p.grad addresses: %s (%d of %d differ)' % (bool((tab['g_ptr'].cpu() == cur).all()),
      int((tab['g_ptr'].cpu() != cur).sum()), cur.numel()), flush=True)

print('=== B: the same model: eager DDP step from the restored state, then a replay again', flush=True)
load(mg, og, S0)
with torch.cuda.stream(wg._effdet_capture_stream):
    step(wg, og)
torch.cuda.synchronize()
Pe, Ge = flat(mg), flat(mg, grad=True)
print('eager DDP vs twin: params %.3e of update, grads diff %.3e' % (rel(Pe, Pt, upd), float((Ge - Gt).norm())), flush=True)
load(mg, og, S0); g(); torch.cuda.synchronize()
Pg2, Gg2 = flat(mg), flat(mg, grad=True)
print('replay after the eager step vs first replay: params %.3e of update; vs twin %.3e; |g| %.4e' % (rel(Pg2, Pg1, upd), rel(Pg2, Pt, upd), float(Gg2.norm())), flush=True)

print('=== C: plain captured step (no DDP)', flush=True)
mp, wp, op_ = build(False)
for _ in range(2):
    step(wp, op_)
gp = GraphedTrainStep(wp, op_, img, ann, warmup=2)
pool_grads = [p.grad for p in ps(mp)]              # the graph's own gradient tensors (its private pool): rewritten by every replay
names = [k for k, p in mp.named_parameters() if p.requires_grad]
gp(); torch.cuda.synchronize()
S1 = snap(mp, op_)
cl, rl = gp(); torch.cuda.synchronize()
Pp = flat(mp)
Gp = torch.cat([g_.detach().reshape(-1).float() for g_ in pool_grads]).clone()
norm_g = float(op_._table['scratch'][0]); loss_g = (float(cl), float(rl))
load(mt, ot, S1)
ot.zero_grad(set_to_none=True)
clt, rlt = wt([img, ann]); (clt.mean() + rlt.mean()).backward(); ot.step(); torch.cuda.synchronize()
Pt1, Gt1 = flat(mt), flat(mt, grad=True)
norm_t = float(ot._table['scratch'][0]); loss_t = (float(clt), float(rlt))
upd1 = float((Pt1 - S1['p']).norm())
print('update norm %.4e | plain replay vs twin eager: params %.3e of update' % (upd1, rel(Pp, Pt1, upd1)), flush=True)
print('losses replay %s twin %s | gradient norm the optimizer measured: replay %.6e twin %.6e | pool grads vs twin grads: |g_pool| %.4e |g_twin| %.4e |diff| %.4e'
      % (loss_g, loss_t, norm_g, norm_t, float(Gp.norm()), float(Gt1.norm()), float((Gp - Gt1).norm())), flush=True)
rows, o = [], 0
for k, p in zip(names, ps(mp)):
    n = p.numel()
    rows.append((float((Gp[o:o + n] - Gt1[o:o + n]).norm()), k, float(Gt1[o:o + n].norm()), float(Gp[o:o + n].norm()),
                 float((Pp[o:o + n] - Pt1[o:o + n]).norm()), float((Pt1[o:o + n] - S1['p'][o:o + n]).norm()))); o += n
rows.sort(reverse=True)
for d, k, gt, gg_, dp, up in rows[:10]:
    print('  %-56s |g_pool-g_twin| %.3e |g_twin| %.3e |g_pool| %.3e  |P diff| %.3e |update| %.3e' % (k, d, gt, gg_, dp, up), flush=True)
same = sum(1 for r in rows if r[0] == 0.0)
print('tensors with bitwise-equal gradients: %d of %d' % (same, len(rows)), flush=True)
steps_g, steps_t = op_._table['steps'].cpu(), ot._table['steps'].cpu()
print('adam step counters equal: %s (replay %s.. twin %s..)' % (bool((steps_g == steps_t).all()), steps_g[:3].tolist(), steps_t[:3].tolist()), flush=True)
print('moments after the step: exp_avg diff %.3e (norm %.3e), exp_avg_sq diff %.3e (norm %.3e)' % (
    float((op_.exp_avg - ot.exp_avg).norm()), float(ot.exp_avg.norm()), float((op_.exp_avg_sq - ot.exp_avg_sq).norm()), float(ot.exp_avg_sq.norm())), flush=True)
dist.barrier()
os._exit(0)
