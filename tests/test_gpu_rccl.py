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
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tests', 'rccl_worker.py'), out, 'bf16x3'], capture_output=True, text=True,
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
