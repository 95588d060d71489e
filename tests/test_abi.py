"""CPU tier: the C-ABI library builds for gfx950, loads, and exports every symbol include/effdet_hip.h declares
(no compute calls -- there is no GPU here)."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    h = open(os.path.join(ROOT, 'include', 'effdet_hip.h')).read()
    h = re.sub(r'/\*.*?\*/', '', h, flags=re.S)
    return sorted(set(re.findall(r'\b(effdet_[a-z0-9_]+)\s*\(', h)))


def test_library_builds_and_exports_every_declared_symbol():
    from efficientdet.pytorch_amd import build, _lib
    path = build.build(verbose=False)
    assert os.path.exists(path)
    L = _lib.lib()
    declared = _declared()
    assert len(declared) >= 35
    missing = [s for s in declared if not hasattr(L, s)]
    assert not missing, missing
    assert sorted(_lib.SYMBOLS) == declared, set(_lib.SYMBOLS) ^ set(declared)
    assert L.effdet_version().decode().startswith('effdet-hip gfx950')
    # ABI generation: header define == what the library was compiled with == what the ctypes binding was written against
    hdr = int(re.search(r'#define\s+EFFDET_ABI_VERSION\s+(\d+)', open(os.path.join(ROOT, 'include', 'effdet_hip.h')).read()).group(1))
    assert hdr == int(L.effdet_abi_version()) == _lib.ABI_VERSION
    assert int(L.effdet_num_anchors(512, 512)) == 49104 and int(L.effdet_num_anchors(1024, 1024)) == 196416


def test_stale_library_is_refused(tmp_path, monkeypatch):
    """A library with another ABI generation (stale build, foreign EFFDET_HIP_LIB) must not be bound: ctypes would call it with
    shifted arguments."""
    import shutil
    import subprocess
    import pytest
    from efficientdet.pytorch_amd import _lib
    if shutil.which('gcc') is None:
        pytest.skip('no gcc')
    src = tmp_path / 'stale.c'
    src.write_text('int effdet_abi_version(void) { return %d; }\nconst char* effdet_version(void) { return "effdet-hip gfx950 stale"; }\n'
                   % (_lib.ABI_VERSION - 1))
    so = tmp_path / 'libstale.so'
    subprocess.run(['gcc', '-shared', '-fPIC', str(src), '-o', str(so)], check=True)
    monkeypatch.setattr(_lib, '_lib', None); monkeypatch.setattr(_lib, 'LIB_PATH', str(so))
    with pytest.raises(RuntimeError, match='ABI generation'):
        _lib.lib()


def test_struct_layouts_match_header():
    """ctypes mirrors of effdet_seg_t / effdet_conv_t / effdet_wgrad_t have the C sizes (LP64, natural alignment)."""
    import ctypes as C
    from efficientdet.pytorch_amd import _lib
    assert C.sizeof(_lib.Seg) == 4 * 4 + 4 * 8
    assert C.sizeof(_lib.ConvDesc) == 10 * 8 + 15 * 4 + 4 + 10 * C.sizeof(_lib.Seg) + 8 + 8 + 8 + 2 * 10 * 8  # 10 ptrs, 15 ints (+pad), 10 segs, w_image_stride, y_split, range_flag, seg_w[10], seg_shift[10]
    assert C.sizeof(_lib.WgradDesc) == 4 * 8 + 13 * 4 + 4 + 5 * C.sizeof(_lib.Seg)        # 4 ptrs, 13 ints (+pad), 5 segs


def test_ctypes_structs_match_the_header_as_gcc_sees_it(tmp_path):
    """Every descriptor struct of include/effdet_hip.h: sizeof as compiled by gcc from the header == the ctypes mirror."""
    import ctypes as C
    import os
    import shutil
    import subprocess
    from efficientdet.pytorch_amd import _lib
    if shutil.which('gcc') is None:
        import pytest
        pytest.skip('no gcc')
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pairs = [('effdet_seg_t', _lib.Seg), ('effdet_conv_t', _lib.ConvDesc), ('effdet_wgrad_t', _lib.WgradDesc), ('effdet_prep_job_t', _lib.PrepJob),
             ('effdet_unpack_job_t', _lib.UnpackJob), ('effdet_se_param_job_t', _lib.SeParamJob), ('effdet_dw_unpack_job_t', _lib.DwUnpackJob),
             ('effdet_tail_job_t', _lib.TailJob)]
    src = tmp_path / 'sz.c'
    src.write_text('#include <stdio.h>\n#include "effdet_hip.h"\nint main(void){' +
                   ''.join('printf("%%zu\\n", sizeof(%s));' % n for n, _ in pairs) + 'return 0;}\n')
    exe = tmp_path / 'sz'
    subprocess.run(['gcc', '-I', os.path.join(root, 'include'), str(src), '-o', str(exe)], check=True)
    sizes = [int(v) for v in subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split()]
    for (name, ct), sz in zip(pairs, sizes):
        assert C.sizeof(ct) == sz, (name, C.sizeof(ct), sz)
    assert C.sizeof(_lib.TailJob) * 24 + 4 * (1 + 25 + 24) < 4096          # a 24-job batch fits the kernel-argument segment


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from efficientdet.pytorch_amd import _lib
    monkeypatch.setattr(_lib, '_lib', None)
    monkeypatch.setattr(_lib, 'LIB_PATH', str(tmp_path / 'nope.so'))
    import pytest
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        _lib.lib()


def test_no_float_atomics_in_the_kernels():
    """Every reduction of the path is a fixed-order sum of per-workgroup partials (bitwise-reproducible runs,
    tests/test_gpu_determinism.py): the only atomics left in csrc/ act on integers (counters, bit masks, hash slots) or on an
    integer count (loss.hip: num_pos), and the build does not ask for unsafe fp atomics."""
    import glob
    src = os.path.join(ROOT, 'efficientdet', 'pytorch_amd', 'csrc')
    for f in sorted(glob.glob(os.path.join(src, '*.hip')) + glob.glob(os.path.join(src, '*.h'))):
        code = re.sub(r'//.*', '', open(f).read())
        hits = [m.group(0) for m in re.finditer(r'atomicAdd\s*\([^;]*;', code)]
        floaty = [h for h in hits if not re.search(r'kcount|kover_n|nvalid|&cnt|\(int|counter', h)]
        assert not floaty, (os.path.basename(f), floaty)
    assert 'unsafe-fp-atomics' not in open(os.path.join(ROOT, 'efficientdet', 'pytorch_amd', 'build.py')).read()


def integration_md_stub():
    """The python code block of INTEGRATION.md section 2 (the ctypes stub a maintainer would paste), with the library path made absolute."""
    md = open(os.path.join(ROOT, 'INTEGRATION.md')).read()
    sec = md[md.index('## 2. Binding the C ABI'):]
    code = re.search(r'```python\n(.*?)```', sec, flags=re.S).group(1)
    assert 'class Conv(C.Structure)' in code and 'def conv3x3_bias_relu' in code
    return code.replace("'efficientdet/pytorch_amd/libeffdet_hip.so'", repr(os.path.join(ROOT, 'efficientdet', 'pytorch_amd', 'libeffdet_hip.so')))


def test_integration_md_stub_matches_the_header(tmp_path):
    """The DOCUMENTED binding (not _lib.py): its Seg / Conv ctypes structs have the sizes and field offsets gcc gives effdet_seg_t /
    effdet_conv_t, and its ABI generation is the header's."""
    import ctypes as C
    import shutil
    import subprocess
    import pytest
    from efficientdet.pytorch_amd import build
    if shutil.which('gcc') is None:
        pytest.skip('no gcc')
    build.build(verbose=False)
    ns = {}
    exec(integration_md_stub(), ns)                         # defines lib, Seg, Conv, conv3x3_bias_relu (no compute call at import)
    fields = [n for n, _ in ns['Conv']._fields_]
    src = tmp_path / 'off.c'
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "effdet_hip.h"\nint main(void){printf("%zu %zu\\n", sizeof(effdet_seg_t), sizeof(effdet_conv_t));'
                   + ''.join('printf("%%zu\\n", offsetof(effdet_conv_t, %s));' % f for f in fields) + 'return 0;}\n')
    exe = tmp_path / 'off'
    subprocess.run(['gcc', '-I', os.path.join(ROOT, 'include'), str(src), '-o', str(exe)], check=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split()
    assert (C.sizeof(ns['Seg']), C.sizeof(ns['Conv'])) == (int(out[0]), int(out[1]))
    for f, off in zip(fields, out[2:]):
        assert getattr(ns['Conv'], f).offset == int(off), f
