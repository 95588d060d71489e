"""Fused train-step tail: global-norm gradient clipping + AdamW in three HIP launches.

Replaces the pair of the reference's train step (train.py:115-118)

    torch.nn.utils.clip_grad_norm_(model.parameters(), 0.1)
    optimizer.step()                      # torch.optim.AdamW(lr=1e-4), train.py:268

with `ClipAdamW(params, lr=..., max_norm=0.1).step()`: same arithmetic (clip coefficient max_norm / (norm + 1e-6)
clamped to 1; decoupled weight decay; bias-corrected moments), one pass for the norm and one for the update over a
device-resident pointer table instead of ~17 foreach / multi-tensor launches.  Only the gradient pointers change from
step to step (fresh tensors after zero_grad(set_to_none=True)); they are refreshed through a pinned staging buffer.
fp32 parameters on one GPU; moments live in two flat arenas.  There is no CPU fallback."""
import ctypes as C

import numpy as np
import torch

from . import _lib as L
from . import ops as ops_mod


class ClipAdamW(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, max_norm=0.0, write_clipped_grads=False):
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, max_norm=max_norm)
        super().__init__(params, defaults)
        if len(self.param_groups) != 1:
            raise ValueError('ClipAdamW supports a single parameter group (the reference uses one, train.py:268)')
        self.write_clipped_grads = bool(write_clipped_grads)
        self._table = None
        self._step = 0
        self._captured = False               # a step() ran under stream capture (its launch sequence is frozen in a hipGraph)

    # ---- the device tables (built once; parameter and moment storage is stable) ----
    def _build(self):
        ps = [p for p in self.param_groups[0]['params'] if p.requires_grad]
        if not ps:
            raise ValueError('no trainable parameters')
        dev = ps[0].device
        for p in ps:
            if p.dtype != torch.float32 or p.device != dev or not p.is_contiguous() or not p.is_cuda:
                raise ValueError('ClipAdamW needs contiguous fp32 parameters on one GPU')
        chunk = int(L.lib().effdet_opt_chunk())
        n = len(ps)
        numel = [p.numel() for p in ps]
        offs = np.concatenate([[0], np.cumsum([(k + 63) // 64 * 64 for k in numel])]).astype(np.int64)
        self.exp_avg = torch.zeros(int(offs[-1]), dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(int(offs[-1]), dtype=torch.float32, device=dev)
        block_tensor, block_first, nb = [], [], 0
        for i, k in enumerate(numel):
            b = (k + chunk - 1) // chunk
            block_first.append(nb); block_tensor += [i] * b; nb += b
        i64 = lambda a: torch.from_numpy(np.asarray(a, dtype=np.int64)).to(dev)
        i32 = lambda a: torch.from_numpy(np.asarray(a, dtype=np.int32)).to(dev)
        t = {
            'params': ps, 'n': n, 'nblocks': nb,
            'p_ptr': i64([p.data_ptr() for p in ps]),
            'm_ptr': i64([self.exp_avg.data_ptr() + 4 * int(o) for o in offs[:-1]]),
            'v_ptr': i64([self.exp_avg_sq.data_ptr() + 4 * int(o) for o in offs[:-1]]),
            'numel': i64(numel), 'block_tensor': i32(block_tensor), 'block_first': i32(block_first),
            'g_ptr': torch.zeros(n, dtype=torch.int64, device=dev),
            # pinned staging ring for the gradient-pointer upload: a slot is rewritten only after the event recorded behind
            # its previous H2D copy has completed (the host may run a step ahead of the stream)
            'g_host': [torch.zeros(n, dtype=torch.int64).pin_memory() for _ in range(2)], 'g_evt': [None, None], 'g_slot': 0,
            'g_graph_host': torch.zeros(n, dtype=torch.int64).pin_memory(),      # source of the memcpy node of a captured step

            'g_last': None,
            'scratch': torch.zeros(64 + nb, dtype=torch.float32, device=dev), 'offs': offs,
            'steps': torch.zeros(n, dtype=torch.int32, device=dev), 'p_sig': [p.data_ptr() for p in ps],
            # hyper-parameters live on the device (read by the update kernel at run time): a captured step follows the
            # schedule of param_groups[0] -- sync_hyper() refreshes them ahead of a graph replay
            'hyper': torch.zeros(6, dtype=torch.float32, device=dev), 'hyper_host': torch.zeros(6, dtype=torch.float32).pin_memory(),
            'hyper_last': None, 'hyper_evt': None,
        }
        for i, p in enumerate(ps):                                     # per-parameter views for state_dict()
            self.state[p] = {'step': torch.tensor(0.0),                # refreshed from the device counters in state_dict()
                             'exp_avg': self.exp_avg[int(offs[i]):int(offs[i]) + numel[i]].view_as(p),
                             'exp_avg_sq': self.exp_avg_sq[int(offs[i]):int(offs[i]) + numel[i]].view_as(p)}
        self._table = t

    # ---- checkpointing (train.py:279-291 saves only the model; resuming needs the moments + per-tensor step counters) ----
    def state_dict(self):
        """torch.optim.AdamW-compatible: per-parameter {'step', 'exp_avg', 'exp_avg_sq'}; 'step' is read back from the
        device-resident counters (they advance on the GPU), the moments are views of the two arenas."""
        if self._table is not None:
            steps = self._table['steps'].cpu()
            for i, p in enumerate(self._table['params']):
                self.state[p]['step'] = torch.tensor(float(steps[i]))
        return super().state_dict()

    def load_state_dict(self, state_dict):
        """Loads a state_dict of this class or of torch.optim.AdamW over the same parameter list: moments are copied INTO
        the arenas and the step counters into the device array (super() alone would leave self.state pointing at fresh
        tensors the kernels never see)."""
        super().load_state_dict(state_dict)
        loaded = {p: dict(st) for p, st in self.state.items()}
        self.state.clear()
        self._build()
        t = self._table
        steps = torch.zeros(t['n'], dtype=torch.int32)
        for i, p in enumerate(t['params']):
            st = loaded.get(p)
            if not st:
                continue
            self.state[p]['exp_avg'].copy_(st['exp_avg'].to(p.device, torch.float32).view_as(p))
            self.state[p]['exp_avg_sq'].copy_(st['exp_avg_sq'].to(p.device, torch.float32).view_as(p))
            steps[i] = int(float(st['step']))
        t['steps'].copy_(steps)
        self._step = int(steps.max()) if t['n'] else 0

    def _hyper_values(self):
        g = self.param_groups[0]
        b1, b2 = g['betas']
        return (float(g['max_norm'] or 0.0), float(g['lr']), float(b1), float(b2), float(g['eps']), float(g['weight_decay']))

    def sync_hyper(self):
        """Upload param_groups[0]'s {max_norm, lr, betas, eps, weight_decay} to the device buffer the update kernel reads, if
        they changed since the last upload.  step() calls it; graph.GraphedTrainStep calls it before every replay, so an lr
        scheduler (train.py:133,269: ReduceLROnPlateau) drives captured steps exactly like eager ones."""
        if self._table is None:
            return
        t, hv = self._table, self._hyper_values()
        if hv == t['hyper_last']:
            return
        if t['hyper_last'] is not None and (hv[0] > 0.0) != (t['hyper_last'][0] > 0.0) and self._captured:
            raise RuntimeError('ClipAdamW: max_norm switched between 0 and > 0 after the step was captured as a hipGraph '
                               '(the norm pass is part of the captured launch sequence); re-capture the step')
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError('ClipAdamW: hyper-parameters changed under stream capture; call sync_hyper() before capturing')
        if t['hyper_evt'] is not None:
            t['hyper_evt'].synchronize()                               # the copy that last read the pinned buffer has finished
        t['hyper_host'].copy_(torch.tensor(hv, dtype=torch.float32))
        t['hyper'].copy_(t['hyper_host'], non_blocking=True)
        ev = torch.cuda.Event(); ev.record()
        t['hyper_evt'], t['hyper_last'] = ev, hv

    @torch.no_grad()
    def step(self, closure=None):
        if len(self.param_groups) != 1:         # (before anything is uploaded or counted: a refused step leaves no trace)
            raise RuntimeError('ClipAdamW honours ONE parameter group (the reference builds one, train.py:266); got %d -- other '
                               "groups' lr / weight_decay would be silently ignored" % len(self.param_groups))
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        if self._table is None or self._table['p_sig'] != [p.data_ptr() for p in self._table['params']]:
            self._build()
        t, g = self._table, self.param_groups[0]
        ptrs = []
        for p in t['params']:
            gr = p.grad
            if gr is None:
                ptrs.append(0)
                continue
            if gr.dtype != torch.float32 or not gr.is_contiguous():
                gr = gr.float().contiguous(); p.grad = gr
            ptrs.append(gr.data_ptr())
        if ptrs != t['g_last']:                                        # (DDP bucket views keep their addresses: no upload)
            if torch.cuda.is_current_stream_capturing():
                # hipGraph capture: the memcpy node keeps reading this pinned buffer on every replay, so it gets a buffer of
                # its own that is not rewritten by eager steps (gradient addresses are static inside the graph's memory pool)
                t['g_graph_host'].copy_(torch.tensor(ptrs, dtype=torch.int64))      # (allocated in _build: no pinned allocation under capture)
                t['g_ptr'].copy_(t['g_graph_host'], non_blocking=True)
                t['g_last'] = None                                     # eager steps after the capture upload again
            else:
                slot = t['g_slot']; t['g_slot'] = slot ^ 1
                if t['g_evt'][slot] is not None:
                    t['g_evt'][slot].synchronize()                     # the copy that last read this pinned slot has finished
                t['g_host'][slot].copy_(torch.tensor(ptrs, dtype=torch.int64))
                t['g_ptr'].copy_(t['g_host'][slot], non_blocking=True)
                ev = torch.cuda.Event(); ev.record()
                t['g_evt'][slot] = ev
                t['g_last'] = ptrs
        self._step += 1
        b1, b2 = g['betas']
        if torch.cuda.is_current_stream_capturing():
            self._captured = True
            if t['hyper_last'] is None:
                raise RuntimeError('ClipAdamW: run one eager step (or sync_hyper()) before capturing the step as a hipGraph')
            if self._hyper_values() != t['hyper_last']:
                raise RuntimeError('ClipAdamW: param_groups changed since the last upload of the device hyper-parameter buffer and '
                                   'the stream is capturing (no upload possible); call sync_hyper() before capturing')
        else:
            self.sync_hyper()
        ops_mod.bump_param_generation()      # parameters are rewritten through raw pointers: Tensor._version does not move
        L.check(L.lib().effdet_clip_adamw_step(L.ptr(t['p_ptr']), L.ptr(t['g_ptr']), L.ptr(t['m_ptr']), L.ptr(t['v_ptr']), L.ptr(t['numel']),
                                               L.ptr(t['block_tensor']), L.ptr(t['block_first']), t['n'], t['nblocks'],
                                               L.ptr(t['scratch']), L.ptr(t['steps']), C.c_float(g['max_norm'] or 0.0),
                                               C.c_float(g['lr']), C.c_float(b1), C.c_float(b2), C.c_float(g['eps']),
                                               C.c_float(g['weight_decay']), int(self.write_clipped_grads), L.ptr(t['hyper']), L.stream_ptr()),
                'effdet_clip_adamw_step')
        return loss

    def grad_norm(self):
        """Total gradient norm measured by the last step() (device scalar; clipping enabled only)."""
        return self._table['scratch'][0] if self._table is not None else None
