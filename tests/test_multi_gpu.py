"""Several GPUs behind one handle (mckpp_hip_multi_*), on the one GPU of the test box: every shard is a context on
device 0, which exercises the sharding, the gather (peer-copy path taken as device-to-device copies, one stream per
shard), the per-shard records and the per-shard restart files exactly as N devices would.  Everything is compared
with the single-context path bit for bit (which the other tests compare with the oracle)."""
import numpy as np
import pytest

import common as cm

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mk(built):
    import torch   # before the library: both bring a HIP runtime, the process must end up with one

    if not torch.cuda.is_available():
        pytest.fail("GPU tests need an MI355X (no HIP device visible)")
    import mckpp_f90_amd as m

    m.load_library()
    return m


STATE = ("U", "X", "Us", "Xs", "hmixd", "hmix", "kmix", "Tref", "uref", "vref", "Ssurf", "old", "new_", "reset_flag")
DIAG = ("rho", "cp", "buoy", "difm", "difs", "dift", "ghat", "wU", "wX", "wXNT", "Rig", "dbloc", "Shsq", "swfrac", "swdk_opt")


def _series(ncol, nrec, seed):
    rng = np.random.default_rng(seed)
    s = np.empty((nrec, 8, ncol))
    for r in range(nrec):
        day = max(0.0, np.sin(2 * np.pi * (2 * r + 3) / 24.0))
        s[r] = [rng.uniform(-0.2, 0.3, ncol), rng.uniform(-0.1, 0.1, ncol), 800.0 * day * np.ones(ncol),
                rng.uniform(-80, -20, ncol), rng.uniform(-300, 0, ncol), rng.uniform(-40, 10, ncol),
                rng.uniform(0, 1e-4, ncol), np.zeros(ncol)]
    return s


@pytest.mark.parametrize("nz,shards", [(40, 3), (69, 2)])
def test_forced_loop_windows_restart_through_the_multi_handle(mk, nz, shards, tmp_path):
    """The reference's forced time loop (src/mckpp_ocean_model_3D.F90:38-58), the output windows
    (src/mckpp_xios_io.F90:74-210) and the restart set (:368-465) for all shards at once:
    multi_set_flux_series / run_forced / window_* / save_restart / load_restart / download(all fields) equal the
    single context; restart files of another shard count or land mask are refused before anything is replaced."""
    ncol, ndtocn, nsteps = 701, 2, 5
    grid = "stretched" if nz == 69 else "uniform"
    series = _series(ncol, (nsteps + 3 + ndtocn - 1) // ndtocn + 1, 11)
    fields = [mk.api.OUT["T"], mk.api.OUT["hmix"], mk.api.OUT["dift"], mk.api.OUT["S"]]

    def drive(h, k3):
        h.upload(k3)
        h.init_ocean(0)
        h.set_flux_series(0, series)
        h.window_select(fields)
        for nt in range(1, nsteps + 1):
            h.run_forced(nt, 1, ndtocn)
            h.window_accumulate()
        out = {}
        for f in fields:
            shape = (ncol,) if f == mk.api.OUT["hmix"] else (ncol, nz + 1)
            for op in range(4):
                a = np.full(shape, -7.0, order="F")
                h.window_fetch(f, op, a)
                out[(f, op)] = a
        return out

    kc1, k31 = cm.make_hip_case(ncol, nz, grid=grid, land_every=5)
    one = mk.MckppHip(kc1)
    w1 = drive(one, k31)
    kc2, k3m = cm.make_hip_case(ncol, nz, grid=grid, land_every=5)
    m = mk.MckppHipMulti(kc2, [0] * shards)
    wm = drive(m, k3m)
    land = k31.run_physics == 0
    for key in w1:
        assert np.array_equal(w1[key], wm[key]), f"window field {key}"
        assert np.all(wm[key][land] == -7.0)
    one.download(k31)
    m.download(k3m)
    for n in STATE + DIAG:
        assert np.array_equal(getattr(k31, n), getattr(k3m, n)), f"download of {n}"

    # restart: save from the multi handle, go on; a fresh handle loads and goes on the same way
    path = tmp_path / "rs"
    m.save_restart(path)
    for h, k3 in ((one, k31), (m, k3m)):
        h.run_forced(nsteps + 1, 3, ndtocn)
        h.download(k3, mk.api.F_RESTART)
    for n in STATE:
        assert np.array_equal(getattr(k31, n), getattr(k3m, n)), f"three more steps: {n}"
    kc3, k3r = cm.make_hip_case(ncol, nz, grid=grid, land_every=5)
    r = mk.MckppHipMulti(kc3, [0] * shards)
    r.upload(k3r)
    r.load_restart(path)
    r.set_flux_series(0, series)
    r.run_forced(nsteps + 1, 3, ndtocn)
    r.download(k3r, mk.api.F_RESTART)
    for n in STATE:
        assert np.array_equal(getattr(k3r, n), getattr(k3m, n)), f"after the restart: {n}"
    # files of another shard count / another land mask: refused, the resident state stays usable
    kc4, k3o = cm.make_hip_case(ncol, nz, grid=grid, land_every=5)
    o = mk.MckppHipMulti(kc4, [0] * (shards + 1))
    o.upload(k3o)
    with pytest.raises(mk.MckppHipError):
        o.load_restart(path)
    kc5, k3l = cm.make_hip_case(ncol, nz, grid=grid, land_every=7)
    l = mk.MckppHipMulti(kc5, [0] * shards)
    l.upload(k3l)
    with pytest.raises(mk.MckppHipError, match="does not belong"):
        l.load_restart(path)
    l.init_ocean(0)
    l.synchronize()
    for h in (one, m, r, o, l):
        h.close()


def test_transfers_with_and_without_pinned_host_arrays(mk, monkeypatch):
    """Upload / download go through pinned caller arrays (hipHostRegister, double-buffered) by default and through
    pageable memory with MCKPP_HIP_NO_HOST_REGISTER=1 or for small arrays: same bytes either way, also after the
    arrays were released and for a second set of arrays on the same context."""
    ncol, nz = 3001, 60          # (ncol x nzp1 doubles) > 1 MiB: the arrays get pinned
    kc, k3 = cm.make_hip_case(ncol, nz, land_every=11)
    ctx = mk.MckppHip(kc)
    ctx.upload(k3)
    ctx.init_ocean(0)
    cm.set_forcing_3d(k3, cm.synth.forcing(ncol, "bench"))
    ctx.set_forcing(k3.sflux)
    ctx.step(1, 2)
    ctx.download(k3)
    kcb, k3b = cm.make_hip_case(ncol, nz, land_every=11)   # a second set of host arrays, same context
    ctx.download(k3b)
    for n in STATE + DIAG:
        assert np.array_equal(getattr(k3, n), getattr(k3b, n)), n
    ctx.release_host_arrays()
    kcc, k3c = cm.make_hip_case(ncol, nz, land_every=11)
    ctx.download(k3c)
    for n in STATE + DIAG:
        assert np.array_equal(getattr(k3, n), getattr(k3c, n)), n
    ctx.close()
