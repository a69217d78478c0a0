#!/usr/bin/env python3
"""Generate the committed golden vectors from the COMPILED REFERENCE.

Run in the build container (needs /root/reference and `make -C oracle ref`):
    python tests/golden/make_golden.py

Produces (inputs + expected outputs only, no reference source):
  eos_ref.npz   - mckpp_abk80 / mckpp_cpsw of the reference
                  (src/mckpp_physics_state_equations.F90) on 4096 (S,T,P) points
  z121_ref.npz  - mckpp_physics_verticalmixing_z121 of the reference
                  (src/mckpp_physics_verticalmixing_z121_mod.F90) on 64 vectors
The rest of the path cannot be built here without a stand-in netcdf module
(DESIGN.md "Oracle pinning"), so there are no reference vectors for it.
"""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import orc  # noqa: E402

dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))  # noqa: E731


def main():
    assert orc.have_ref(), "oracle/_ref/libmckpp_ref.so missing (make -C oracle ref)"
    R = orc.ref()
    rng = np.random.default_rng(20261003)
    n = 4096
    s = rng.uniform(0.0, 42.0, n)
    t = rng.uniform(-4.0, 35.0, n)       # includes T < -2 (clamped in the reference)
    p = rng.uniform(0.05, 6000.0, n)
    # include the model's own operating points: fresh water / brine at the surface
    s[:8] = [0.0, 4.0, 35.0, 35.0, 40.0, 0.0, 4.0, 34.5]
    t[:8] = [10.0, 10.0, 15.0, -3.0, 0.0, 28.0, 28.0, 2.0]
    p[:8] = [2.5, 2.5, 1.6667, 100.0, 1000.0, 1.0, 1.0, 200.0]
    alpha, beta, sig0, sig, cp = (np.zeros(n) for _ in range(5))
    R.ref_abk80_batch(n, dp(s), dp(t), dp(p), dp(alpha), dp(beta), dp(sig0), dp(sig))
    R.ref_cpsw_batch(n, dp(s), dp(t), dp(p), dp(cp))
    np.savez_compressed(os.path.join(HERE, "eos_ref.npz"), s=s, t=t, p=p, alpha=alpha, beta=beta,
                        sig0=sig0, sig=sig, cp=cp)
    vin, vout, wout, kms = [], [], [], []
    for i in range(64):
        km = int(rng.integers(3, 110))
        v = rng.uniform(-0.6, 1.6, km + 2)
        if i % 4 == 0:
            v[rng.integers(1, km + 1, 3)] = [0.0, 0.8, -1e-30]   # edge values of the weight test
        w = rng.uniform(0, 1, km + 2)
        vi = np.zeros(112); vi[:km + 2] = v
        R.ref_z121(km + 1, 0.0, 0.8, dp(v), dp(w))
        vo = np.zeros(112); vo[:km + 2] = v
        wo = np.zeros(112); wo[:km + 2] = w
        vin.append(vi); vout.append(vo); wout.append(wo); kms.append(km)
    np.savez_compressed(os.path.join(HERE, "z121_ref.npz"), km=np.array(kms), vin=np.array(vin),
                        vout=np.array(vout), wout=np.array(wout))
    print("wrote eos_ref.npz, z121_ref.npz")


if __name__ == "__main__":
    main()
