#!/usr/bin/env python3
"""Golden vector of the round-5 part of the compiler-convention probe: what amdflang (-fdefault-real-8 -O2, the
flags of oracle/Makefile) returns for EXP, SQRT, ABS, x**2, the Jerlov transmission forms of swfrac / swdk,
SIGN, MAX / MIN / AMAX1 / AMIN1 with two to four arguments (equal and signed-zero operands included) and
ifix / int / float (oracle/conv_probe.F90 - our own source, no reference code).  Run in the build container:
    python tests/golden/make_conv_golden2.py
"""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from mckpp_f90_amd import synth  # noqa: E402
from oracle import orc  # noqa: E402

dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))  # noqa: E731
ip = lambda a: a.ctypes.data_as(C.POINTER(C.c_int))  # noqa: E731


def model_depths():
    """dm(0:nz), -zm(1:nzp1) of every BASELINE grid and a sweep of boundary-layer depths: the arguments the model
    itself hands to swfrac / swdk."""
    z = [np.linspace(0.1, 1000.0, 3000)]
    for nz, grid in ((40, "uniform"), (60, "uniform"), (100, "uniform"), (69, "stretched")):
        zm, hm, dm = synth.uniform_grid(nz) if grid == "uniform" else synth.stretched_grid(nz)
        z += [dm, -zm[1:nz + 2]]
    return np.ascontiguousarray(np.concatenate(z))


def inputs(seed=20261005, n=12000):
    rng = np.random.default_rng(seed)
    ja1, ja2 = (0.35, 0.6, 1.0, 1.5, 1.4), (23.0, 20.0, 17.0, 14.0, 7.9)
    zs = model_depths()
    own = np.concatenate([-zs / a for a in ja1 + ja2])
    x = np.concatenate([rng.uniform(-80.0, 0.0, n), own[own > -800.0], rng.uniform(-1.0, 1.0, 2000), 10 ** rng.uniform(-12, 2, 2000),
                        np.array([0.0, -0.0, -80.0, -745.0, -746.0, -708.4, 1.0, 709.0, 710.0, 1e-300, -1e-300, 0.54, 4.6])])
    m = 6000
    a = np.concatenate([rng.normal(0, 1, m), 10 ** rng.uniform(-20, 3, m) * rng.choice([-1.0, 1.0], m)])
    b = np.concatenate([rng.normal(0, 1, m), 10 ** rng.uniform(-20, 3, m) * rng.choice([-1.0, 1.0], m)])
    c = np.concatenate([rng.normal(0, 1, m), 10 ** rng.uniform(-20, 3, m) * rng.choice([-1.0, 1.0], m)])
    d = np.concatenate([rng.normal(0, 1, m), 10 ** rng.uniform(-20, 3, m) * rng.choice([-1.0, 1.0], m)])
    eq = rng.random(2 * m) < 0.15          # equal operands
    b[eq] = a[eq]
    eq = rng.random(2 * m) < 0.1
    c[eq] = b[eq]
    sp = np.array([0.0, -0.0, 1e-16, -1e-16, 1e-17, -1e-17, -2e-16, 5e-324, -5e-324, 0.5, -0.5, 1.0, 0.8, 0.1, -80.0])
    A, B = np.meshgrid(sp, sp)             # every pair of the special values, signed zeros included
    a = np.concatenate([a, A.ravel()]); b = np.concatenate([b, B.ravel()])
    c = np.concatenate([c, B.ravel()[::-1]]); d = np.concatenate([d, A.ravel()[::-1]])
    xc = np.concatenate([rng.uniform(0.0, 892.0, 4000), rng.uniform(-2.0, 2.0, 1000),
                         np.array([0.0, -0.0, 1.0, 0.5, 0.9999999999999999, 1.0 - 1e-20, 1e-20, -0.5, -1.0, 889.99999, 890.0, 891.5, 1e6,
                                   48.0, 48.999999999, 49.0])])
    return {"x": np.ascontiguousarray(x), "a": np.ascontiguousarray(a), "b": np.ascontiguousarray(b),
            "c": np.ascontiguousarray(c), "d": np.ascontiguousarray(d), "xc": np.ascontiguousarray(xc), "z": zs}


def run_probe(P, inp):
    """amdflang's results for the inputs, as a dict of arrays."""
    out = {}
    x = inp["x"]; n = len(x)
    e, s, a, q = (np.zeros(n) for _ in range(4))
    P.conv_probe_unary(n, dp(x), dp(e), dp(s), dp(a), dp(q))
    out.update(exp=e, sqrtabs=s, abs=a, sq=q)
    z = inp["z"]; nzv = len(z)
    L = orc.lib()
    for j in range(1, 6):
        jr = np.zeros(3); L.orc_conv_jerlov(j, dp(jr))
        for fact, zz, tag in ((-1.0, z, "hbl"), (1.0, -z, "zm")):      # swfrac(-1, hbl, j) | swfrac_opt: fact = hbf = 1, z = zm(l) < 0
            zz = np.ascontiguousarray(zz)
            sw, sk = np.zeros(nzv), np.zeros(nzv)
            P.conv_probe_swfrac(nzv, dp(zz), fact, jr[0], jr[1], jr[2], dp(sw), dp(sk))
            out[f"swfrac_{tag}_{j}"] = sw
            if tag == "zm":
                out[f"swdk_{j}"] = sk                                   # swdk(-dm(k), j): negative argument
    a_, b_, c_, d_ = inp["a"], inp["b"], inp["c"], inp["d"]; m = len(a_)
    names = ("sign", "sign_half", "sign_half_eps", "max", "min", "amax1", "amin1", "min3", "min4", "max3")
    o = [np.zeros(m) for _ in names]
    P.conv_probe_binary(m, dp(a_), dp(b_), dp(c_), dp(d_), *[dp(v) for v in o])
    out.update(dict(zip(names, o)))
    xc = inp["xc"]; k = len(xc)
    ifx, itr, icl = (np.zeros(k, dtype=np.int32) for _ in range(3)); fl = np.zeros(k)
    P.conv_probe_casts(k, dp(xc), ip(ifx), ip(itr), ip(icl), dp(fl))
    out.update(ifix_eps=ifx, int=itr, int_clamped=icl, frac=fl)
    return out


def run_oracle(inp):
    """The oracle's C lowering of the same constructs on the same inputs."""
    L = orc.lib()
    out = {}
    x = inp["x"]; n = len(x)
    e, s, a, q = (np.zeros(n) for _ in range(4))
    L.orc_conv_unary(n, dp(x), dp(e), dp(s), dp(a), dp(q))
    out.update(exp=e, sqrtabs=s, abs=a, sq=q)
    z = inp["z"]; nzv = len(z)
    for j in range(1, 6):
        for fact, zz, tag in ((-1.0, z, "hbl"), (1.0, -z, "zm")):
            zz = np.ascontiguousarray(zz)
            sw, sk = np.zeros(nzv), np.zeros(nzv)
            L.orc_conv_swfrac(nzv, dp(zz), fact, j, dp(sw), dp(sk))
            out[f"swfrac_{tag}_{j}"] = sw
            if tag == "zm":
                out[f"swdk_{j}"] = sk
    a_, b_, c_, d_ = inp["a"], inp["b"], inp["c"], inp["d"]; m = len(a_)
    names = ("sign", "sign_half", "sign_half_eps", "max", "min", "amax1", "amin1", "min3", "min4", "max3")
    o = [np.zeros(m) for _ in names]
    L.orc_conv_binary(m, dp(a_), dp(b_), dp(c_), dp(d_), *[dp(v) for v in o])
    out.update(dict(zip(names, o)))
    xc = inp["xc"]; k = len(xc)
    ifx, itr, icl = (np.zeros(k, dtype=np.int32) for _ in range(3)); fl = np.zeros(k)
    L.orc_conv_casts(k, dp(xc), ip(ifx), ip(itr), ip(icl), dp(fl))
    out.update(ifix_eps=ifx, int=itr, int_clamped=icl, frac=fl)
    return out


def main():
    P = orc.conv_probe()
    assert P is not None and hasattr(P, "conv_probe_unary"), "amdflang not available"
    inp = inputs()
    got = run_probe(P, inp)
    np.savez_compressed(os.path.join(HERE, "conv_probe_intrinsics.npz"), **{"in_" + k: v for k, v in inp.items()},
                        **{"out_" + k: v for k, v in got.items()})
    print("wrote conv_probe_intrinsics.npz:", {k: len(v) for k, v in inp.items()})


if __name__ == "__main__":
    main()
