"""The Fortran host layer (mckpp_f90_amd/fortran): modules with the reference's
names over the C-ABI, exercised by kpp_driver, a forced run shaped like the
reference's main loop.  The GPU test feeds it the same bits as the C-ABI tests
and requires bit-identical results; the CPU test checks that it builds and
refuses to run without a device."""
import os
import subprocess

import numpy as np
import pytest

import common as cm

DRIVER = os.path.join(cm.ROOT, "mckpp_f90_amd", "kpp_driver")


def _write_case(path, kc, k3, sf6, nsteps, use_1d, flags=0, shards=0):
    npts, nz = k3.npts, kc.nz
    with open(path, "wb") as f:
        np.array([npts, nz, nsteps, use_1d, kc.nztmax, flags, shards, 1 if shards else 0], dtype=np.int32).tofile(f)
        np.array([kc.dto]).tofile(f)
        for a in (kc.zm, kc.hm, kc.dm):
            np.asarray(a, dtype=np.float64).tofile(f)
        for a in (k3.U, k3.X):
            np.asarray(a).ravel(order="F").tofile(f)
        for a in (k3.f, k3.Sref, k3.SSref, k3.Ssurf, k3.ocdepth):
            np.asarray(a).tofile(f)
        np.asarray(k3.jerlov, dtype=np.int32).tofile(f)
        np.asarray(k3.run_physics, dtype=np.float64).tofile(f)
        np.asarray(sf6).ravel(order="F").tofile(f)


def _read_out(path, kc, npts, vmix=False, gather=False):
    nz, nzp1, nzt = kc.nz, kc.nzp1, kc.nztmax
    out = {}
    with open(path, "rb") as f:
        def rd(shape, dt=np.float64):
            n = int(np.prod(shape))
            return np.fromfile(f, dtype=dt, count=n).reshape(shape, order="F")
        out["U"] = rd((npts, nzp1, 2)); out["X"] = rd((npts, nzp1, 2))
        out["Us"] = rd((npts, nzp1, 2, 2)); out["Xs"] = rd((npts, nzp1, 2, 2))
        out["hmix"] = rd((npts,)); out["kmix"] = rd((npts,)); out["hmixd"] = rd((npts, 2))
        out["Tref"] = rd((npts,)); out["Ssurf"] = rd((npts,))
        out["old"] = rd((npts,), np.int32); out["new_"] = rd((npts,), np.int32)
        out["difm"] = rd((npts, nzt + 1)); out["ghat"] = rd((npts, nzt)); out["rho"] = rd((npts, nzt + 2))
        if gather:
            out["g_hmix"] = rd((npts,)); out["g_T"] = rd((npts, nzp1))
        if vmix:
            out["vm_h"] = rd((npts,)); out["vm_k"] = rd((npts,))
            out["vm_difm"] = rd((npts, nzt + 1)); out["vm_difs"] = rd((npts, nzt + 1)); out["vm_dift"] = rd((npts, nzt + 1))
            out["vm_ghat"] = rd((npts, nzt))
    return out


def test_fortran_layer_builds_and_fails_loudly_without_gpu(built, tmp_path):
    import torch

    assert os.path.exists(DRIVER), "kpp_driver not built (make -C mckpp_f90_amd/fortran)"
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    kc, k3 = cm.make_hip_case(8, 40)
    _write_case(tmp_path / "case.bin", kc, k3, cm.synth.forcing(8), 1, 0)
    r = subprocess.run([DRIVER, str(tmp_path / "case.bin"), str(tmp_path / "out.bin")], capture_output=True, text=True)
    assert r.returncode != 0 and "MCKPP-HIP ERROR" in r.stderr
    assert not os.path.exists(tmp_path / "out.bin")


@pytest.mark.gpu
@pytest.mark.parametrize("ncol,nz,nsteps,use_1d,land,flags", [(500, 60, 3, 0, 0, 0), (96, 40, 2, 0, 4, 0), (12, 40, 2, 1, 5, 0),
                                                              (96, 40, 2, 0, 4, 64)])
def test_fortran_driver_matches_cabi_path(built, tmp_path, ncol, nz, nsteps, use_1d, land, flags):
    """flags 0: the session's default - every mckpp_physics_driver call leaves ALL of kpp_3d_fields current (the
    reference's contract, src/mckpp_types_transfer.F90:199-327; kpp_driver stops if anything is stale); flags 64:
    the opt-in scalar-group download with mckpp_hip_sync_host before the output."""
    import mckpp_f90_amd as mk

    kc, k3 = cm.make_hip_case(ncol, nz, land_every=land)
    sf = cm.synth.forcing(ncol, "bench")
    _write_case(tmp_path / "case.bin", kc, k3, sf, nsteps, use_1d, flags=flags)
    r = subprocess.run([DRIVER, str(tmp_path / "case.bin"), str(tmp_path / "out.bin")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr + r.stdout
    got = _read_out(tmp_path / "out.bin", kc, ncol)
    mk.mckpp_initialize_ocean_model(k3, kc)
    cm.set_forcing_3d(k3, sf)
    for nt in range(1, nsteps + 1):
        mk.mckpp_physics_driver(k3, kc, nt)
    for n in ("U", "X", "Us", "Xs", "hmix", "kmix", "hmixd", "Tref", "Ssurf", "old", "new_", "difm", "ghat", "rho"):
        assert np.array_equal(got[n], getattr(k3, n)), n


@pytest.mark.gpu
@pytest.mark.parametrize("use_1d", [0, 1])
def test_fortran_layer_writes_the_located_warnings(built, tmp_path, use_1d):
    """The reference names the column - longitude, latitude - in a warning on stderr when it iterates beyond
    itermax+1 passes (src/mckpp_physics_ocnstep_mod.F90:184-191; mckpp_print_warning,
    src/mckpp_log_messages.F90:52-63).  With itermax = 4 (driver flag 128) the steps from the analytic start flag
    columns (tests/test_parity_gpu.py::test_long_iteration_status_on_device): the Fortran layer must write one
    warning per flagged column and step, with that column's location, through mckpp_physics_driver and through the
    one-column mckpp_physics_ocnstep alike; the results stay those of the C-ABI path."""
    import re

    import mckpp_f90_amd as mk

    ncol, nz, nsteps = (120, 40, 3) if not use_1d else (24, 40, 3)
    kc, k3 = cm.make_hip_case(ncol, nz)
    sf = cm.synth.forcing(ncol, "bench")
    _write_case(tmp_path / "case.bin", kc, k3, sf, nsteps, use_1d, flags=128)
    r = subprocess.run([DRIVER, str(tmp_path / "case.bin"), str(tmp_path / "out.bin")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr + r.stdout
    got = _read_out(tmp_path / "out.bin", kc, ncol)
    kc.itermax = 4
    ctx = mk.mckpp_initialize_ocean_model(k3, kc)
    cm.set_forcing_3d(k3, sf)
    want = []   # (step, point) of every flagged column
    for nt in range(1, nsteps + 1):
        mk.mckpp_physics_driver(k3, kc, nt)
        st, nf, npass = ctx.status()
        want += [(nt, int(i) + 1) for i in np.nonzero(st & 2)[0]]
    assert want, "no column ran beyond itermax+1 passes"
    for n in ("U", "X", "hmix", "kmix", "Tref"):
        assert np.array_equal(got[n], getattr(k3, n)), n
    lines = [ln.strip() for ln in r.stderr.splitlines()]
    seen = []
    for i, ln in enumerate(lines):
        m = re.match(r"long iteration at timestep\s+(\d+)\s+location = \(\s*([-0-9.Ee+]+)\s*,\s*([-0-9.Ee+]+)\s*\)", ln)
        if m:
            assert lines[i - 1] == "Warning in MCKPP_PHYSICS_OCNSTEP:"
            ipt = round(float(m.group(2)) / 0.5)
            assert abs(float(m.group(3)) - (-60 + 0.25 * ipt)) < 1e-9
            assert re.search(r"passes =\s+\d+", lines[i + 2]) and re.search(rf"ipt =\s+{ipt}$", lines[i + 2])
            seen.append((int(m.group(1)), ipt))
    assert seen == want


@pytest.mark.gpu
@pytest.mark.parametrize("flags", [1, 2, 3])
def test_fortran_fluxes_and_bottomtemp_wrappers(built, tmp_path, flags):
    """mckpp_fluxes (constant forcing, L_FLUXDATA=.F., src/mckpp_fluxes_mod.F90:41-49) every step and
    the L_VARY_BOTTOM_TEMP override, through the Fortran modules, against the same calls on the C-ABI."""
    import mckpp_f90_amd as mk

    ncol, nz, nsteps = 77, 40, 3
    kc, k3 = cm.make_hip_case(ncol, nz, land_every=6)
    sf = cm.synth.forcing(ncol, "bench")
    _write_case(tmp_path / "case.bin", kc, k3, sf, nsteps, 0, flags)
    r = subprocess.run([DRIVER, str(tmp_path / "case.bin"), str(tmp_path / "out.bin")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr + r.stdout
    got = _read_out(tmp_path / "out.bin", kc, ncol)
    if flags & 2:
        kc.L_VARY_BOTTOM_TEMP = 1
        k3.bottom_temp[:] = np.asarray(k3.X[:, nz, 0]) + 0.125
    ctx = mk.mckpp_initialize_ocean_model(k3, kc)
    cm.set_forcing_3d(k3, sf)
    one = np.ones(ncol)
    for nt in range(1, nsteps + 1):
        if flags & 1:
            ctx.fluxes(nt, 0.01 * one, 0 * one, 200 * one, 0 * one, -150 * one, 0 * one, 6e-5 * one, 0 * one)
        mk.mckpp_physics_driver(k3, kc, nt, new_forcing=not (flags & 1))
    for n in ("U", "X", "Us", "Xs", "hmix", "kmix", "hmixd", "Tref", "Ssurf", "old", "new_", "difm", "ghat", "rho"):
        assert np.array_equal(got[n], getattr(k3, n)), n
    if flags & 2:
        act = np.nonzero(k3.run_physics)[0]
        assert np.array_equal(got["X"][act, nz, 0], k3.bottom_temp[act])


@pytest.mark.gpu
@pytest.mark.parametrize("ncol,nz,land", [(40, 40, 0), (30, 69, 4)])
def test_fortran_verticalmixing_wrapper(built, tmp_path, ncol, nz, land):
    """mckpp_physics_verticalmixing(kpp_1d_fields, kpp_const_fields, hmixn, kmixn) - the third signature
    of the reference's call surface (src/mckpp_physics_verticalmixing_mod.F90:14) - called by kpp_driver
    for every column of its final state: hmixn, kmixn and the mixing coefficients equal the oracle's vmix
    on the same state."""
    from oracle import orc

    nsteps = 2
    kc, k3 = cm.make_hip_case(ncol, nz, land_every=land)
    sf = cm.synth.forcing(ncol, "bench")
    _write_case(tmp_path / "case.bin", kc, k3, sf, nsteps, 0, flags=4)
    r = subprocess.run([DRIVER, str(tmp_path / "case.bin"), str(tmp_path / "out.bin")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr + r.stdout
    got = _read_out(tmp_path / "out.bin", kc, ncol, vmix=True)
    oc, ob = cm.make_oracle(ncol, nz, init=True, exp_mode=1)
    for nt in range(1, nsteps + 1):
        orc.physics_driver(oc, ob, nt)
    act = np.nonzero(k3.run_physics)[0]
    assert np.array_equal(got["hmix"][act], ob["hmix"][act])
    orc.vmix_only(oc, ob, nsteps)
    assert np.array_equal(got["vm_h"][act], ob["hmix"][act])
    assert np.array_equal(got["vm_k"][act], ob["kmix"][act])
    for n, o in (("vm_difm", "difm"), ("vm_difs", "difs"), ("vm_dift", "dift")):
        assert np.array_equal(got[n][act][:, :nz + 2], ob[o][act][:, :nz + 2]), n
    assert np.array_equal(got["vm_ghat"][act][:, :nz], ob["ghat"][act][:, 1:nz + 1])


@pytest.mark.gpu
@pytest.mark.parametrize("ncol,nz,shards,land", [(301, 60, 3, 5), (200, 69, 2, 0)])
def test_fortran_driver_on_several_device_shards(built, tmp_path, ncol, nz, shards, land):
    """kpp_driver with mckpp_hip_ndevices = 2 or 3 (all shards on device 0: the box has one GPU): the
    drop-in mckpp_physics_driver drives every shard from the one Fortran process, and the output gather
    (mckpp_hip_gather_field -> mckpp_hip_multi_gather) returns the same hmix and T as the download."""
    import mckpp_f90_amd as mk

    nsteps = 3
    kc, k3 = cm.make_hip_case(ncol, nz, land_every=land)
    sf = cm.synth.forcing(ncol, "bench")
    _write_case(tmp_path / "case.bin", kc, k3, sf, nsteps, 0, flags=8, shards=shards)
    r = subprocess.run([DRIVER, str(tmp_path / "case.bin"), str(tmp_path / "out.bin")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr + r.stdout
    got = _read_out(tmp_path / "out.bin", kc, ncol, gather=True)
    mk.mckpp_initialize_ocean_model(k3, kc)
    cm.set_forcing_3d(k3, sf)
    for nt in range(1, nsteps + 1):
        mk.mckpp_physics_driver(k3, kc, nt)
    for n in ("U", "X", "Us", "Xs", "hmix", "kmix", "hmixd", "Tref", "Ssurf", "old", "new_", "difm", "ghat", "rho"):
        assert np.array_equal(got[n], getattr(k3, n)), n
    ocean = k3.run_physics != 0
    assert np.array_equal(got["g_hmix"][ocean], k3.hmix[ocean]) and np.all(got["g_hmix"][~ocean] == -1)
    assert np.array_equal(got["g_T"][ocean], k3.X[ocean, :, 0]) and np.all(got["g_T"][~ocean] == -1)


@pytest.mark.gpu
@pytest.mark.parametrize("flags,shards,nz", [(16, 0, 40), (16, 3, 60), (32, 2, 69), (32 + 64, 2, 60)])
def test_fortran_forced_run_and_output_windows_on_all_devices(built, tmp_path, flags, shards, nz):
    """kpp_driver with its time loop on the devices: mckpp_hip_all_set_flux_series + mckpp_hip_all_run_forced
    (the reference's loop, src/mckpp_ocean_model_3D.F90:38-58, one call for all steps and all shards), and - flag
    32 - step by step with an output window on every shard (mean hmix, maximum T fetched through the gather).
    The forced run leaves everything on the devices and mckpp_hip_sync_host brings it back before the driver
    writes its output; flag 64 is the session's opt-in to a scalar-group-only per-step download.  Against mckpp_fluxes + mckpp_physics_driver per step on the C-ABI."""
    import mckpp_f90_amd as mk

    ncol, nsteps = 211, 4
    kc, k3 = cm.make_hip_case(ncol, nz, land_every=6)
    sf = cm.synth.forcing(ncol, "bench")
    _write_case(tmp_path / "case.bin", kc, k3, sf, nsteps, 0, flags=flags, shards=shards)
    r = subprocess.run([DRIVER, str(tmp_path / "case.bin"), str(tmp_path / "out.bin")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr + r.stdout
    got = _read_out(tmp_path / "out.bin", kc, ncol, gather=bool(flags & 32))
    ctx = mk.mckpp_initialize_ocean_model(k3, kc)
    cm.set_forcing_3d(k3, sf)
    one = np.ones(ncol)
    hm, tmax = np.zeros(ncol), None
    for nt in range(1, nsteps + 1):
        ctx.fluxes(nt, 0.01 * one, 0 * one, 200 * one, 0 * one, -150 * one, 0 * one, 6e-5 * one, 0 * one)
        mk.mckpp_physics_driver(k3, kc, nt, new_forcing=False)
        hm = hm + k3.hmix
        tmax = k3.X[:, :, 0].copy() if tmax is None else np.maximum(tmax, k3.X[:, :, 0])
    for n in ("U", "X", "Us", "Xs", "hmix", "kmix", "hmixd", "Tref", "Ssurf", "old", "new_", "difm", "ghat", "rho"):
        assert np.array_equal(got[n], getattr(k3, n)), n
    if flags & 32:
        ocean = k3.run_physics != 0
        assert np.array_equal(got["g_hmix"][ocean], (hm / nsteps)[ocean]) and np.all(got["g_hmix"][~ocean] == -1)
        assert np.array_equal(got["g_T"][ocean], tmax[ocean]) and np.all(got["g_T"][~ocean] == -1)
