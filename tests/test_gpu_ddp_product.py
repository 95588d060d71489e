"""N>1 on the PRODUCT path: two ranks of the HIP model under ddp.wrap + ClipAdamW on the one visible GPU.
(tests/test_ddp_gloo.py covers the sharding / wrapping host logic on CPU with the oracle as the module.)"""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(('127.0.0.1', 0)); p = s.getsockname()[1]; s.close()
    return p


@pytest.mark.parametrize('dtype,wrap', [('f32', 'ddp.wrap'), ('bf16', 'ddp.wrap'), ('f32', 'reference')])
def test_two_rank_ddp_hip_model(tmp_path, dtype, wrap):
    """wrap = 'reference': the reference's own DistributedDataParallel(model, device_ids=[gpu], find_unused_parameters=True)
    (train.py:250-251) with the five never-executed tensors left trainable, instead of ddp.wrap."""
    out = str(tmp_path / 'ddp.json')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(_free_port()), os.path.join(ROOT, 'tests', 'ddp_product_worker.py'), out, dtype, wrap]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0', OMP_NUM_THREADS='4')
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    log_dir = os.path.join(ROOT, 'gpurun_out')
    if os.path.isdir(log_dir):                                      # keep the log (scratch on the GPU box, merged back)
        open(os.path.join(log_dir, 'ddp_product_%s_%s.log' % (dtype, wrap)), 'w').write(r.stdout[-4000:] + '\n--- stderr ---\n' + r.stderr[-4000:])
    assert r.returncode == 0, r.stderr[-3000:]
    res = json.load(open(out))
    if os.path.isdir(log_dir):
        json.dump(res, open(os.path.join(log_dir, 'ddp_product_%s_%s.json' % (dtype, wrap)), 'w'))
    assert res['world'] == 2 and res['grad_tensors'] > 250
    # DDP average of the two shard gradients == full-batch gradient (fp32: the batch is summed in another order, measured 1.2e-6)
    assert res['grad_worst_rel'] <= (5e-3 if dtype == 'f32' else 0.1), res
    assert res['grad_worst_l2'] <= (5e-3 if dtype == 'f32' else 0.1), res
    assert res['finite'] and res['prep_replay']
    assert res['param_checksums'][0] == res['param_checksums'][1], res       # replicas bit-identical after 3 ClipAdamW steps
