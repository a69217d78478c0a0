import os, sys
import numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import common as cm
import mckpp_f90_amd as mk
ncol, nz = int(os.environ.get("DBG_NCOL", 9000)), 60
def run(multi, nsteps, env={}):
    for k,v in env.items(): os.environ[k]=v
    os.environ["MCKPP_MULTISTEP"] = multi
    kc, k3 = cm.make_hip_case(ncol, nz)
    ctx = mk.MckppHip(kc); ctx.upload(k3); ctx.init_ocean(0)
    cm.set_forcing_3d(k3, cm.synth.forcing(ncol, "bench")); ctx.set_forcing(k3.sflux)
    ctx.step(1, nsteps); ctx.download(k3)
    st, nf, npass = ctx.status(); ctx.close()
    for k in env: os.environ.pop(k)
    return np.asarray(k3.U).copy(), npass.copy()
ref = {ns: run("0", ns, {"MCKPP_SOLO": "0"}) for ns in (3,)}
def trial(tag, multi, ns, env):
    u, n = run(multi, ns, env)
    bad = np.flatnonzero((u != ref[ns][0]).any(axis=(1, 2)))
    print(f"{tag}: {len(bad)} columns differ {bad[:8]}", flush=True)
for rep in range(2):
    trial("multi default", "1", 3, {})
    trial("multi KMAX=1", "1", 3, {"MCKPP_VIEW_KMAX": "1"})
    trial("multi AFTER=1e6 (sticky only)", "1", 3, {"MCKPP_SOLO_AFTER": "1000000"})
    trial("multi LIMIT=1e6", "1", 3, {"MCKPP_SOLO_LIMIT": "1000000"})
    trial("multi LIMIT=1e6 AFTER=0", "1", 3, {"MCKPP_SOLO_LIMIT": "1000000", "MCKPP_SOLO_AFTER": "0"})
    trial("single default", "0", 3, {})
    trial("single LIMIT=1e6 AFTER=0", "0", 3, {"MCKPP_SOLO_LIMIT": "1000000", "MCKPP_SOLO_AFTER": "0"})
    trial("multi SOLO=0", "1", 3, {"MCKPP_SOLO": "0"})
