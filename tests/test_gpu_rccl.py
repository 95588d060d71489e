"""RCCL on the hardware a test box has: a world_size = 1 'nccl' process group on the one GPU runs RCCL
communicator bring-up, ddp.wrap's bucket hooks + RCCL all-reduce, and the captured DDP step (collectives inside the hipGraph).
The 2-rank control flow is covered over gloo (tests/test_gpu_ddp_product.py, tests/test_ddp_gloo.py); this is the leg that puts
the real collective library under the product path.  Reference path: train.py:237-258."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(('127.0.0.1', 0)); p = s.getsockname()[1]; s.close()
    return p


@pytest.fixture(scope='module')
def rccl_run(tmp_path_factory):
    out = str(tmp_path_factory.mktemp('rccl') / 'rccl.json')
    env = dict(os.environ, RANK='0', LOCAL_RANK='0', WORLD_SIZE='1', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(_free_port()),
               HSA_ENABLE_IPC_MODE_LEGACY='0', OMP_NUM_THREADS='4')
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tests', 'rccl_worker.py'), out, 'f32_hf16x3_bwd_bf16x3'], capture_output=True, text=True,   # (the bench headline arithmetic)
                       timeout=600, env=env, cwd=ROOT)
    res = json.load(open(out)) if os.path.exists(out) else {'stage': 'none'}
    log_dir = os.path.join(ROOT, 'gpurun_out')
    if os.path.isdir(log_dir):                                      # keep the log (scratch on the GPU box, merged back)
        open(os.path.join(log_dir, 'rccl_world1.log'), 'w').write(r.stdout[-4000:] + '\n--- stderr ---\n' + r.stderr[-6000:])
        json.dump(res, open(os.path.join(log_dir, 'rccl_world1.json'), 'w'))
    return r, res


def test_rccl_collective_and_eager_ddp_equal_the_bare_module(rccl_run):
    r, res = rccl_run
    assert res['stage'] in ('eager', 'capture', 'captured', 'checked', 'done'), (res, r.stderr[-3000:])
    assert res['allreduce_ok'] and res['rccl_version'][0] >= 2
    assert res['finite'] and res['prep_replay']
    assert res['eager_losses_ddp'] == res['eager_losses_plain'], res          # same kernels, same masks: identical floats
    assert res['eager_bitwise'], res                                          # parameters after 3 ClipAdamW steps bit for bit
    assert res['bucket_views'], res                                           # gradient pointers stayed put: no re-upload per step


def test_ddp_step_with_rccl_all_reduce_captured_as_hipgraph(rccl_run):
    r, res = rccl_run
    assert r.returncode == 0 and res['stage'] == 'done', (res, r.stderr[-3000:])
    le, lg = res['eager_tail_losses'], res['graph_losses']
    assert all(np.isfinite(lg)) and res['graph_finite']
    for a, b in zip(le, lg):
        assert abs(a - b) <= 1e-5 * abs(a), (le, lg)               # replay k == eager step 11 + k: same kernels, same masks
    assert len(set(lg)) == len(lg)                                  # replays do real, different steps
    # parameters after 14 steps: eager vs 11 warm-ups + 3 replays (the gate of test_graphed_train_step_tracks_eager: AdamW sign noise)
    assert res['graph_param_diff_mean'] < 0.5 * res['lr'] and res['graph_param_diff_max'] <= res['steps'] * 2 * res['lr'] + 1e-6, res
    assert np.isfinite(res['graph_second_batch_loss']) and res['graph_second_batch_loss'] != lg[-1]
    # the strict form: one replay == one eager DDP step from the same state (parameters compared against the step's own update)
    rv = res['replay_vs_eager']
    assert rv['finite'] and rv['eager_vs_eager'] == 0.0 and rv['replay_vs_replay'] == 0.0 and rv['replay_vs_eager'] <= 1e-6, rv


def test_capture_probe_answers_in_child_processes():
    """bench.py's pre-flight (ddp.rccl_graph_probe): the single-rank form, and the multi-rank form's rendezvous (the ranks' children meet
    on a port of their own, RANK / WORLD_SIZE taken from the caller) exercised with the one rank this box has."""
    sys.path.insert(0, ROOT)
    from efficientdet.pytorch_amd import ddp
    ok, note = ddp.rccl_graph_probe(0)
    assert ok and 'world_size-1' in note, note
    ok, note = ddp.rccl_graph_probe(0, rank=0, world_size=1, port=ddp.probe_port(_free_port()))
    assert ok, note
    # a rank whose peers never show up must come back with a verdict, not hang the job: world_size 2 with nobody at rank 1
    ok, note = ddp.rccl_graph_probe(0, timeout=20, rank=0, world_size=2, port=_free_port())
    assert not ok and 'timed out' in note, note


@pytest.mark.timeout(1200)
def test_bench_two_rccl_ranks_real_or_refused_never_hung(tmp_path):
    """`python bench.py --gpus 2` through the self-launch + supervisor + child legs with the REAL RCCL backend.
    On a box with >= 2 GPUs this is the multi-rank run itself: one JSON line, n_gpus 2, and -- when the captured DDP step was the path --
    its replay-vs-eager self-check on both ranks.  On a 1-GPU box (EFFDET_BENCH_SHARE_GPU=1: both ranks on device 0) RCCL refuses the
    communicator ("Duplicate GPU detected"): both attempts must then fail FAST with the failure reported and a non-zero exit code -- no
    hang, no number."""
    import time
    import torch
    two = torch.cuda.device_count() >= 2
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0', OMP_NUM_THREADS='4')
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_PORT', 'MASTER_ADDR'):
        env.pop(k, None)
    if not two:
        env['EFFDET_BENCH_SHARE_GPU'] = '1'
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--batch', '4', '--steps', '3', '--warmup', '1', '--no-roofline'],
                       capture_output=True, text=True, timeout=1100, env=env, cwd=ROOT)
    dt = time.time() - t0
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith('{')]
    if two:
        assert r.returncode == 0 and len(lines) == 1, r.stderr[-3000:]
        out = lines[0]
        assert out['n_gpus'] == 2 and out['value'] > 0 and out['config']['global_batch'] == 8
        note = out['config']['ddp_graph']
        if note['captured']:
            sc = note['self_check']
            assert sc['finite'] and sc['ranks_hold_equal_parameters'] and sc['replay_vs_eager_rel_to_update'] <= 1e-3, sc
    else:
        assert r.returncode != 0 and not lines, (r.returncode, r.stdout[-500:])
        assert 'every attempt failed' in r.stderr and 'attempt 2' in r.stderr
        assert dt < 200, dt
