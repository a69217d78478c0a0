"""CPU tests of the drop-in boundary: the library loads, exports every symbol
declared in include/mckpp_hip.h, the ctypes mirror of the structs matches the
C layout, host helpers agree with the oracle, and compute entry points fail
loudly (no CPU fallback) when no gfx950 device is present."""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest

import common as cm
from oracle import orc

ROOT = cm.ROOT
HEADER = os.path.join(ROOT, "include", "mckpp_hip.h")


def _declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(mckpp_h(?:ip|ost)_[a-z_0-9]+)\s*\(", src)))


def test_library_exports_every_declared_symbol(built):
    import mckpp_f90_amd as mk

    lib = mk.load_library()
    names = _declared_functions()
    assert len(names) >= 18
    for n in names:
        assert hasattr(lib, n), f"{n} declared in mckpp_hip.h but not exported"


def test_struct_layout_matches_header(built, tmp_path):
    """sizeof/offsetof from a C compile of the header vs the ctypes mirror in api.py."""
    from mckpp_f90_amd import api

    csrc = tmp_path / "layout.c"
    fields_c = [f[0] for f in api._ConstC._fields_]
    fields_s = [f[0] for f in api._StateC._fields_]
    body = ['#include <stdio.h>', '#include <stddef.h>', f'#include "{HEADER}"', "int main(void){",
            'printf("%zu %zu\\n", sizeof(mckpp_const_c), sizeof(mckpp_state_ptrs_c));']
    for f in fields_c:
        body.append(f'printf("c {f} %zu\\n", offsetof(mckpp_const_c, {f}));')
    for f in fields_s:
        body.append(f'printf("s {f} %zu\\n", offsetof(mckpp_state_ptrs_c, {f}));')
    body.append("return 0;}")
    csrc.write_text("\n".join(body))
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", str(csrc), "-o", str(exe)])
    out = subprocess.check_output([str(exe)]).decode().split("\n")
    sz = out[0].split()
    assert int(sz[0]) == C.sizeof(api._ConstC) and int(sz[1]) == C.sizeof(api._StateC)
    for line in out[1:]:
        if not line:
            continue
        which, name, off = line.split()
        cls = api._ConstC if which == "c" else api._StateC
        assert getattr(cls, name).offset == int(off), (which, name)


def test_host_lookup_and_tri_match_oracle(built):
    import mckpp_f90_amd as mk

    nz = 60
    kc = mk.KppConstFields(nz)
    mk.mckpp_physics_lookup(kc)
    oc = orc.Const(nz)
    # Fortran wmt(0:891,0:49) == oracle [j, i] view transposed
    assert np.array_equal(np.asarray(kc.wmt).T, oc.wmt)
    assert np.array_equal(np.asarray(kc.wst).T, oc.wst)
    assert np.array_equal(kc.tri[0:nz + 1, 0, 0], oc.tri0[0:nz + 1])
    assert np.array_equal(kc.tri[0:nz + 1, 1, 0], oc.tri1[0:nz + 1])
    assert np.array_equal(kc.zm, oc.zm[1:nz + 2]) and np.array_equal(kc.dm, oc.dm[0:nz + 1])


def test_no_cpu_fallback(built):
    """Without a HIP device every compute entry point must fail with an error, never compute."""
    import torch

    import mckpp_f90_amd as mk

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    kc = mk.KppConstFields(40)
    mk.mckpp_physics_lookup(kc)
    with pytest.raises(mk.MckppHipError):
        mk.MckppHip(kc)
    k3 = mk.Kpp3dFields(4, kc)
    with pytest.raises(mk.MckppHipError):
        mk.mckpp_physics_driver(k3, kc, 1)


def test_missing_library_fails_loudly(built, tmp_path, monkeypatch):
    import mckpp_f90_amd as mk

    monkeypatch.setattr(mk, "_lib", None)
    monkeypatch.setattr(mk, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        mk.load_library()


def test_product_does_not_touch_oracle():
    """The package and the C sources never import, include or link the oracle."""
    pkg = os.path.join(ROOT, "mckpp_f90_amd")
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".cpp", ".hip", ".h", ".F90", ".f90", "Makefile")):
                txt = open(os.path.join(dirpath, fn), errors="ignore").read()
                assert "liboracle" not in txt and "mckpp_oracle" not in txt and "from oracle" not in txt \
                    and "import oracle" not in txt, os.path.join(dirpath, fn)


def test_synthetic_generators_are_sliceable():
    a = cm.synth.columns(30, 12)
    idx = np.arange(1, 30, 3)
    b = cm.synth.columns(len(idx), 12, index=idx, ntotal=30)
    for k in ("T", "S", "U", "f", "Sref"):
        assert np.array_equal(a[k][idx], b[k])
    assert np.array_equal(cm.synth.forcing(30)[idx], cm.synth.forcing(len(idx), index=idx))


def test_shard_mask_logic(built):
    """mckpp_host_shard_mask (the column-to-device map of mckpp_hip_multi_upload): the j-th run_physics
    point goes to shard j mod ndev - shards are disjoint, cover the ocean, differ by at most one column."""
    import mckpp_f90_amd as mk

    rng = np.random.default_rng(3)
    for npts, ndev in [(1, 1), (17, 3), (1000, 8), (1000, 7), (5, 8)]:
        rp = (rng.uniform(size=npts) > 0.35).astype(np.int32)
        total = np.zeros(npts, dtype=np.int32)
        counts = []
        ocean = np.flatnonzero(rp)
        for d in range(ndev):
            m, n = mk.host_shard_mask(rp, ndev, d)
            assert n == int(m.sum())
            assert np.array_equal(np.flatnonzero(m), ocean[d::ndev])      # round-robin in ipt order
            total += m
            counts.append(n)
        assert np.array_equal(total, rp) and max(counts) - min(counts) <= 1
    with pytest.raises(mk.MckppHipError):
        mk.host_shard_mask(np.ones(4, dtype=np.int32), 2, 2)
