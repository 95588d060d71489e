"""Build libeffdet_hip.so (hand-written HIP for gfx950) in-tree with hipcc.

No CMake, no torch extension machinery: the library is torch-free (pure C ABI, raw pointers) and is
bound with ctypes (``_lib.py``).  Objects go to ``csrc/build/`` (git-ignored), the shared library
next to this file so that it travels with the tree to the GPU box.
"""
import concurrent.futures as cf
import glob
import hashlib
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB = os.path.join(HERE, 'libeffdet_hip.so')
PER_FILE = {'postprocess.hip': ['-ffp-contract=off']}
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wall',
         '-Wno-unused-function'] + os.environ.get('EFFDET_HIPCC_EXTRA', '').split()     # e.g. -DEFFDET_WGRAD_TR_WAVES=4 for A/B runs


def _hipcc():
    for c in ('hipcc', '/opt/rocm/bin/hipcc'):
        p = shutil.which(c)
        if p:
            return p
    raise RuntimeError('hipcc not found: the HIP extension cannot be built')


def _digest(paths):
    h = hashlib.sha256()
    for p in sorted(paths):
        h.update(p.encode()); h.update(open(p, 'rb').read())
    h.update(' '.join(FLAGS).encode())
    return h.hexdigest()


def build(force=False, verbose=True):
    srcs = sorted(glob.glob(os.path.join(CSRC, '*.hip')))
    deps = srcs + glob.glob(os.path.join(CSRC, '*.h')) + [os.path.join(HERE, '..', '..', 'include', 'effdet_hip.h')]
    deps = [os.path.abspath(d) for d in deps]
    stamp = os.path.join(CSRC, 'build', 'stamp')
    dig = _digest(deps)
    if not force and os.path.exists(LIB) and os.path.exists(stamp) and open(stamp).read() == dig:
        return LIB
    os.makedirs(os.path.join(CSRC, 'build'), exist_ok=True)
    hipcc = _hipcc()

    def one(src):
        obj = os.path.join(CSRC, 'build', os.path.basename(src)[:-4] + '.o')
        cmd = [hipcc] + FLAGS + PER_FILE.get(os.path.basename(src), []) + ['-c', src, '-o', obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError('hipcc failed for %s:\n%s' % (src, r.stderr))
        if verbose and r.stderr.strip():
            sys.stderr.write(r.stderr)
        return obj

    with cf.ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        objs = list(ex.map(one, srcs))
    r = subprocess.run([hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError('link failed:\n' + r.stderr)
    open(stamp, 'w').write(dig)
    if verbose:
        print('built', LIB)
    return LIB


if __name__ == '__main__':
    build(force='--force' in sys.argv)
