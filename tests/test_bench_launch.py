"""CPU tier: bench.py's N > 1 control flow without a GPU (EFFDET_BENCH_DRYRUN=1: the legs' child processes do the process-group
plumbing only and report a result marked dry_run).  What runs for real: the self-launch under torch.distributed.run when WORLD_SIZE is
unset (the reference self-spawns too, train.py:311-326), the supervisor ranks (gloo), one child process per rank and attempt on a port
of its own, the all-rank verdict, the fall-back from the captured-DDP path to eager DDP when a rank's child dies, the exit codes."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, timeout=300, **env):
    e = dict(os.environ, EFFDET_BENCH_DRYRUN='1', OMP_NUM_THREADS='1', **env)
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_PORT', 'MASTER_ADDR'):
        e.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + args, capture_output=True, text=True, timeout=timeout, env=e, cwd=ROOT)
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    return r, [json.loads(l) for l in lines]


@pytest.mark.timeout(600)
def test_bench_launches_itself_for_n_gpus():
    r, lines = _run(['--gpus', '2', '--steps', '2', '--warmup', '1'])
    assert r.returncode == 0, r.stderr[-3000:]
    assert len(lines) == 1, r.stdout                      # ONE line, from rank 0
    out = lines[0]
    assert out['dry_run'] and out['n_gpus'] == 2
    # first attempt = the captured DDP step, and it was accepted by every rank
    assert [h['ok_on_every_rank'] for h in out['attempts']] == [True] and 'captured' in out['attempts'][0]['path']


@pytest.mark.timeout(600)
def test_failed_capture_attempt_falls_back_to_eager_ddp_in_fresh_processes():
    r, lines = _run(['--gpus', '2', '--steps', '2', '--warmup', '1'], EFFDET_BENCH_FAIL_LEG='graph')
    assert r.returncode == 0, r.stderr[-3000:]
    out = lines[0]
    assert [h['ok_on_every_rank'] for h in out['attempts']] == [False, True]
    assert 'captured' in out['attempts'][0]['path'] and 'eager' in out['attempts'][1]['path'] and out['path'] == 'eager'
    assert 'attempt 1' in r.stderr and 'failed' in r.stderr          # the failure is reported, not hidden
    # the rank whose child survived was stopped through the store instead of waiting out the timeout
    assert out['attempts'][0]['seconds'] < 120


@pytest.mark.timeout(600)
def test_every_path_failing_is_an_error_exit_not_a_number():
    r, lines = _run(['--gpus', '2', '--no-ddp-graph'], EFFDET_BENCH_FAIL_LEG='eager')
    assert r.returncode != 0 and not lines
    assert 'every attempt failed' in r.stderr


def test_world_size_mismatch_is_reported_not_asserted():
    e = dict(os.environ, WORLD_SIZE='4', RANK='0', LOCAL_RANK='0')
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2'], capture_output=True, text=True, timeout=120, env=e, cwd=ROOT)
    assert r.returncode == 2 and 'launches them itself' in r.stderr and 'Traceback' not in r.stderr
