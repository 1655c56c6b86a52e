"""The C-ABI library loads and exports every symbol include/panst3r_hip.h declares (no compute calls: no GPU here)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, 'include', 'panst3r_hip.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(pst_[a-z0-9_]+)\s*\(', text)))


def test_library_builds_and_exports_all_symbols():
    from panst3r_amd.build import build
    from panst3r_amd import hip
    path = build(verbose=False)
    lib = ctypes.CDLL(path)
    syms = declared_symbols()
    assert len(syms) >= 15
    for s in syms:
        assert hasattr(lib, s), s
    assert set(syms) == set(hip.EXPORTS)
    lib.pst_abi_version.restype = ctypes.c_int
    assert lib.pst_abi_version() == hip.ABI_VERSION


def test_struct_layouts_match_header():
    """ctypes mirrors of pst_gemm_params / pst_attn_params have the field order of the header."""
    from panst3r_amd import hip
    text = open(os.path.join(ROOT, 'include', 'panst3r_hip.h')).read()
    for cname, ctype in (('pst_gemm_params', hip.GemmParams), ('pst_attn_params', hip.AttnParams)):
        body = re.search(r'typedef struct %s \{(.*?)\} %s;' % (cname, cname), text, flags=re.S).group(1)
        body = re.sub(r'/\*.*?\*/', '', body, flags=re.S)
        names = []
        for decl in body.split(';'):
            decl = decl.strip()
            if not decl:
                continue
            parts = decl.split(',')
            first = re.findall(r'[A-Za-z_][A-Za-z0-9_]*', parts[0])[-1]
            names.append(first)
            names += [re.findall(r'[A-Za-z_][A-Za-z0-9_]*', p)[-1] for p in parts[1:]]
        assert names == [f[0] for f in ctype._fields_], (cname, names)


def test_product_refuses_cpu_tensors():
    import pytest
    import torch
    from panst3r_amd import hip
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        hip.gemm(torch.zeros(64, 64, dtype=torch.bfloat16), torch.zeros(64, 64, dtype=torch.bfloat16), torch.zeros(64, 64))


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, 'panst3r_amd')
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith('.py'):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r'^\s*(from|import)\s+oracle\b', src, flags=re.M), f


def test_only_the_allowed_places_import_the_oracle():
    """oracle/ is test infrastructure: besides tests/ only __graft_entry__.smoke() and bench.py's cpu_baseline leg may import it.
    tools/ must not (oracle-checking diagnostics live in tests/diag/), and bench.py only inside cpu_baseline()."""
    pat = re.compile(r'^(\s*)(from|import)\s+oracle\b', flags=re.M)
    for f in os.listdir(os.path.join(ROOT, 'tools')):
        if f.endswith('.py'):
            assert not pat.search(open(os.path.join(ROOT, 'tools', f)).read()), f
    src = open(os.path.join(ROOT, 'bench.py')).read()
    hits = [m for m in pat.finditer(src)]
    assert hits and all(len(m.group(1)) > 0 for m in hits)                      # function-level imports only
    # ... and only inside the functions of the cpu_baseline leg (cpu_baseline, cpu_c1): find the enclosing def of every import
    for m in hits:
        defs = [d for d in re.finditer(r'^def (\w+)\(', src[:m.start()], flags=re.M)]
        assert defs and defs[-1].group(1).startswith('cpu_'), defs[-1].group(1) if defs else None
    # (__graft_entry__.build() import-checks the Python oracle as its "build the checker" step, which is allowed; smoke() uses it)


def test_diag_and_tool_scripts_compile():
    """tests/diag, tools/ and the golden generator only run by hand (GPU box / build container): keep them at least syntactically valid."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, 'tests', 'diag', '*.py')) + glob.glob(os.path.join(ROOT, 'tools', '*.py')) +
                   [os.path.join(ROOT, 'tests', 'golden', 'make_golden.py'), os.path.join(ROOT, 'bench.py'), os.path.join(ROOT, '__graft_entry__.py')])
    assert len(files) >= 12
    for f in files:
        compile(open(f).read(), f, 'exec')            # syntax only: nothing is executed or written
