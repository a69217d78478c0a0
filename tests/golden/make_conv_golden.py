#!/usr/bin/env python3
"""Golden vector of the compiler-convention probe: inputs and what amdflang (-fdefault-real-8 -O2, the
flags of oracle/Makefile) returns for x**3, x**4, x**(1./2.), x**(1./3.), x**(1./4.) and a few unkinded
literals (oracle/conv_probe.F90 - our own source, no reference code).  Run in the build container:
    python tests/golden/make_conv_golden.py
"""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import orc  # noqa: E402

dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))  # noqa: E731


def inputs():
    rng = np.random.default_rng(20261003)
    n = 4096
    x = np.concatenate([rng.uniform(0, 1e-3, n), rng.uniform(0, 10, n), 10 ** rng.uniform(-12, 3, n),
                        rng.uniform(1, 17, n),                       # 1 - c*zeta of the lookup table
                        np.array([0.0, 1.0, 2.0, 0.04, 4e-7, 5e-324, 1e-300, 1e300 ** 0.25])])
    return np.ascontiguousarray(x)


def main():
    P = orc.conv_probe()
    assert P is not None, "amdflang not available"
    x = inputs()
    n = len(x)
    outs = [np.zeros(n) for _ in range(5)]
    P.conv_probe_powers(n, dp(x), *[dp(o) for o in outs])
    lit = np.zeros(12)
    P.conv_probe_literals(dp(lit))
    np.savez_compressed(os.path.join(HERE, "conv_probe.npz"), x=x, p3=outs[0], p4=outs[1], ph=outs[2], pt=outs[3],
                        pq=outs[4], literals=lit)
    print("wrote conv_probe.npz", n)


if __name__ == "__main__":
    main()
