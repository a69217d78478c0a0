"""the optional physics switches of ocnint / ocnstep / check_profile on the device
(src/mckpp_physics_ocnint_mod.F90:97-215, src/mckpp_physics_solvers.F90:176-335,
src/mckpp_physics_verticalmixing_ddmix_mod.F90, src/mckpp_physics_ocnstep_mod.F90:317-340,
src/mckpp_physics_overrides.F90:42-125).  Every switch combination is compared bit for bit with the
oracle's restatement on identical inputs."""
import numpy as np
import pytest

import common as cm

pytestmark = pytest.mark.gpu

FIELDS = cm.PROFILE_FIELDS + cm.SCALAR_FIELDS + ["hmixd0", "hmixd1"] + list(cm.DIAG_FIELDS.keys()) + cm.EXT_SCALARS


@pytest.fixture(scope="module")
def mk(built):
    import torch

    if not torch.cuda.is_available():
        pytest.fail("GPU tests need an MI355X (no HIP device visible)")
    import mckpp_f90_amd as m

    return m


def _case(mk, ncol, nz, switches, prep, nsteps=2, grid="uniform"):
    """switches: dict of kpp_const switch -> value (same names on both sides);
    prep(k3, ob, col_T, col_S): fill the optional input fields identically on both sides."""
    from oracle import orc

    oc, ob = cm.make_oracle(ncol, nz, init=False, exp_mode=1, grid=grid, **switches)
    kc, k3 = cm.make_hip_case(ncol, nz, grid=grid)
    for k, v in switches.items():
        setattr(kc, k, v)
    prep(k3, ob)
    ctx = mk.mckpp_initialize_ocean_model(k3, kc)
    orc.init_ocean(oc, ob, 0)
    sf = cm.synth.forcing(ncol, "bench")
    ob["sflux"] = sf
    cm.set_forcing_3d(k3, sf)
    for nt in range(1, nsteps + 1):
        ctx = mk.mckpp_physics_driver(k3, kc, nt)
        orc.physics_driver(oc, ob, nt)
        st, nf, npass = ctx.status()
        assert np.array_equal(st, ob["status"]) and np.array_equal(npass, ob["npasses"])
        res = cm.compare(k3, ob, nz, FIELDS)
        bad = {k: v for k, v in res.items() if v[2] != 0}
        assert not bad, f"{switches} step {nt}: {bad}"
    return k3, ob


def _set2(k3, ob, name, arr):
    """same (ncol, nzp1) profile field on both sides"""
    nzp1 = arr.shape[1]
    getattr(k3, name)[:, :] = arr
    ob.a[name][:, 1:nzp1 + 1] = arr


def test_relax_sst_and_calconly(mk):
    def prep(k3, ob):
        n = k3.npts
        r = np.full(n, 1.0 / (5 * 86400.0)); r[::4] = 0.0        # some columns not relaxed
        k3.relax_sst[:] = r; ob["relax_sst"] = r
        sst = k3.X[:, 0, 0] + 1.5
        k3.SST0[:] = sst; ob["SST0"] = sst
    k3, ob = _case(mk, 96, 40, dict(L_RELAX_SST=1), prep)
    assert np.any(k3.fcorr != 0) and np.all(k3.fcorr[::4] == 0)
    _case(mk, 32, 40, dict(L_RELAX_SST=1, L_RELAX_CALCONLY=1), prep)


def test_fcorr_twod_and_withz(mk):
    def prep2(k3, ob):
        v = np.linspace(-80.0, 80.0, k3.npts)
        k3.fcorr_twod[:] = v; ob["fcorr_twod"] = v
    _case(mk, 64, 40, dict(L_FCORR=1), prep2)

    def prepz(k3, ob):
        nzp1 = k3.X.shape[1]
        z = np.arange(nzp1)[None, :]
        _set2(k3, ob, "fcorr_withz", 5.0 * np.exp(-z / 10.0) * np.linspace(-1, 1, k3.npts)[:, None])
        _set2(k3, ob, "sfcorr_withz", 1e-7 * np.cos(z / 7.0) * np.ones((k3.npts, 1)))
    k3, ob = _case(mk, 64, 40, dict(L_FCORR_WITHZ=1, L_SFCORR_WITHZ=1), prepz)
    assert np.any(k3.tinc_fcorr != 0) and np.any(k3.scorr != 0)


def test_relax_ocnt_and_sal(mk):
    def prep(k3, ob):
        n = k3.npts
        _set2(k3, ob, "ocnT_clim", np.asarray(k3.X[:, :, 0]) - 0.3)
        _set2(k3, ob, "sal_clim", np.asarray(k3.X[:, :, 1]) + 0.05)
        r = np.full(n, 1.0 / (30 * 86400.0))
        k3.relax_ocnT[:] = r; ob["relax_ocnT"] = r
        k3.relax_sal[:] = 2 * r; ob["relax_sal"] = 2 * r
    k3, ob = _case(mk, 64, 60, dict(L_RELAX_OCNT=1, L_RELAX_SAL=1), prep)
    assert np.any(k3.ocnTcorr != 0) and np.any(k3.sinc_fcorr != 0)


def test_double_diffusion(mk):
    def prep(k3, ob):
        # salt-fingering favourable stratification on half of the columns: warm salty over cold fresh
        nzp1 = k3.X.shape[1]
        z = np.linspace(0, 1, nzp1)[None, :]
        S = np.asarray(k3.X[:, :, 1]).copy()
        S[::2] = 0.4 - 0.8 * z
        k3.X[:, :, 1] = S
        ob.a["S"][:, 1:nzp1 + 1] = S
    _case(mk, 64, 40, dict(LDD=1), prep, nsteps=3)


def test_double_diffusion_with_the_most_slots_a_workgroup_takes(mk, monkeypatch):
    """21 slots x 3 species = 63 lanes of the manager wave in M3 (hbl, blmix scalars: one lane per (slot, species));
    the launcher's own choice for this depth has fewer"""
    monkeypatch.setenv("MCKPP_PS", "21x4x1")

    def prep(k3, ob):
        nzp1 = k3.X.shape[1]
        z = np.linspace(0, 1, nzp1)[None, :]
        S = np.asarray(k3.X[:, :, 1]).copy()
        S[::2] = 0.4 - 0.8 * z
        k3.X[:, :, 1] = S
        ob.a["S"][:, 1:nzp1 + 1] = S
    _case(mk, 150, 20, dict(LDD=1), prep, nsteps=3)


def test_current_damping(mk):
    _case(mk, 64, 40, dict(L_DAMP_CURR=1, dt_uvdamp=360), lambda k3, ob: None)


def test_no_freeze_and_isotherm_and_clim_reset(mk):
    def prep(k3, ob):
        nzp1 = k3.X.shape[1]
        T = np.asarray(k3.X[:, :, 0]).copy()
        T[::3, :] = -2.2                       # below -1.8 everywhere: clamped, freeze_flag counts levels
        T[1::3, :] = 12.0                      # isothermal columns: reset to climatology
        k3.X[:, :, 0] = T
        ob.a["T"][:, 1:nzp1 + 1] = T
        _set2(k3, ob, "ocnT_clim", 8.0 + 10.0 * np.exp(-np.arange(nzp1) / 15.0)[None, :] * np.ones((k3.npts, 1)))
        _set2(k3, ob, "sal_clim", np.asarray(k3.X[:, :, 1]) * 0.5)
        k3.U[2::3, 0:3, 0] = 40.0              # absurd currents: trap -> climatology + U_init
        ob.a["U"][2::3, 1:4] = 40.0
    sw = dict(L_NO_FREEZE=1, L_NO_ISOTHERM=1, clim_present=1, iso_bot=20, iso_thresh=0.002)
    k3, ob = _case(mk, 60, 40, sw, prep, nsteps=1)
    # frozen columns clamped, isothermal columns reset (negative count), trapped columns reset to climatology
    assert np.any(k3.freeze_flag > 0) and np.any(k3.reset_flag < 0) and np.any(np.abs(k3.reset_flag) == 999)
    _case(mk, 60, 40, sw, prep, nsteps=3)


@pytest.mark.parametrize("nz", [40, 69])
def test_climatology_reset_alone_runs_on_the_default_kernel(mk, nz):
    """The shipped namelist names T and S climatology files (ocnT_file, sal_file /= 'none') and switches nothing
    else on: check_profile then resets a failed column's T and S to the climatology instead of leaving them
    (src/mckpp_physics_overrides.F90:57-78).  That needs the two input fields but none of the optional-physics
    kernel code, so such a context runs the default build."""
    def prep(k3, ob):
        nzp1 = k3.X.shape[1]
        _set2(k3, ob, "ocnT_clim", 8.0 + 10.0 * np.exp(-np.arange(nzp1) / 15.0)[None, :] * np.ones((k3.npts, 1)))
        _set2(k3, ob, "sal_clim", np.asarray(k3.X[:, :, 1]) * 0.5)
        k3.U[2::3, 0:3, 0] = 40.0              # absurd currents: trap, ten retries, reset to climatology + U_init
        ob.a["U"][2::3, 1:4] = 40.0
    k3, ob = _case(mk, 60, nz, dict(clim_present=1), prep, nsteps=1, grid="stretched" if nz == 69 else "uniform")
    clim = 8.0 + 10.0 * np.exp(-np.arange(nz + 1) / 15.0)
    assert np.array_equal(np.asarray(k3.X[2, :, 0]), clim)      # a failed column came back as the climatology
    kc = mk.KppConstFields(nz)
    kc.clim_present = 1
    mk.mckpp_physics_lookup(kc)
    ctx = mk.MckppHip(kc)
    assert ctx.kernel_name == "k_column_ps"
    ctx.close()
    _case(mk, 60, nz, dict(clim_present=1), prep, nsteps=3, grid="stretched" if nz == 69 else "uniform")


@pytest.mark.parametrize("grid", ["uniform", "stretched"])
def test_prescribed_advection_modes(mk, grid):
    def prep(k3, ob):
        n = k3.npts
        for c in range(n):
            mode = 1 + (c % 7)
            k3.nmodeadv[c, 1] = 2
            k3.modeadv[c, 0, 1] = mode
            k3.modeadv[c, 1, 1] = 1 + ((c + 3) % 7)
            k3.advection[c, 0, 1] = 1e-6 * (1 + c % 5)
            k3.advection[c, 1, 1] = -5e-7
        ob["nmodeadv"][:, 1] = k3.nmodeadv[:, 1]
        ob["modeadv"][:, 1, :] = k3.modeadv[:, :, 1]
        ob["advection"][:, 1, :] = k3.advection[:, :, 1]
    nz = 60 if grid == "uniform" else 69
    _case(mk, 70, nz, dict(L_ADVECT=1), prep, grid=grid)


@pytest.mark.parametrize("nz,switches", [(40, {}), (69, {}), (40, dict(L_FCORR_WITHZ=1))])
def test_vary_bottom_temp(mk, nz, switches):
    """L_VARY_BOTTOM_TEMP: mckpp_physics_overrides_bottomtemp after the column loop
    (src/mckpp_physics_driver_mod.F90:67-71, src/mckpp_physics_overrides.F90:12-24), on the default
    kernels (which carry no correction rows of their own) and next to an ocnint correction switch."""
    from oracle import orc

    ncol = 77
    oc, ob = cm.make_oracle(ncol, nz, init=False, exp_mode=1, **switches)
    kc, k3 = cm.make_hip_case(ncol, nz, land_every=6)
    for k, v in switches.items():
        setattr(kc, k, v)
    kc.L_VARY_BOTTOM_TEMP = 1
    if switches:
        z = np.arange(nz + 1)[None, :]
        _set2(k3, ob, "fcorr_withz", 5.0 * np.exp(-z / 10.0) * np.linspace(-1, 1, ncol)[:, None])
    ctx = mk.mckpp_initialize_ocean_model(k3, kc)
    orc.init_ocean(oc, ob, 0)
    active = np.nonzero(k3.run_physics)[0]
    sf = cm.synth.forcing(ncol, "bench")
    ob["sflux"] = sf
    cm.set_forcing_3d(k3, sf)
    for nt in (1, 2, 3):
        bt = np.asarray(k3.X[:, nz, 0]) + 0.25 * np.sin(np.arange(ncol) + nt)
        k3.bottom_temp[:] = bt
        ctx = mk.mckpp_physics_driver(k3, kc, nt)
        orc.physics_driver(oc, ob, nt)
        orc.bottomtemp(oc, ob, bt)
        res = cm.compare(k3, ob, nz, FIELDS, active)
        bad = {k: v for k, v in res.items() if v[2] != 0}
        assert not bad, f"bottomtemp {switches} step {nt}: {bad}"
        assert np.array_equal(k3.X[active, nz, 0], bt[active])
        assert np.any(k3.ocnTcorr[active, nz] != 0)
    land = np.nonzero(k3.run_physics == 0)[0]
    assert np.all(k3.tinc_fcorr[land] == 0)
    ctx.set_diagnostics(0)
    with pytest.raises(mk.MckppHipError):
        ctx.bottomtemp(k3.bottom_temp)


@pytest.mark.parametrize("nz", [30, 69, 100, 150, 200, 509])
def test_options_on_other_depths(mk, nz):
    """The optional physics where a slot's items span one to three wavefronts (k_column_ps<EXT>): the level
    counts of check_profile (damped levels, frozen levels), the isotherm sums and the double-diffusion
    neighbours cross wave boundaries here."""
    nzp1 = nz + 1
    z = np.arange(nzp1)[None, :]

    def check_kernel(k3, ob):
        pass

    # double diffusion: fingering-favourable salinity on half of the columns, all the way down
    def prep_ldd(k3, ob):
        S = np.asarray(k3.X[:, :, 1]).copy()
        S[::2] = 0.4 - 0.8 * np.linspace(0, 1, nzp1)[None, :]
        k3.X[:, :, 1] = S
        ob.a["S"][:, 1:nzp1 + 1] = S
    _case(mk, 40, nz, dict(LDD=1), prep_ldd, nsteps=2)

    # current damping with strong deep currents (levels beyond the first wave are damped too)
    def prep_damp(k3, ob):
        k3.U[::2, min(60, nz - 8):, 0] = 3.0
        k3.U[1::4, min(66, nz - 4):, 1] = -2.5
        ob.a["U"][:, 1:nzp1 + 1] = k3.U[:, :, 0]
        ob.a["V"][:, 1:nzp1 + 1] = k3.U[:, :, 1]
    k3, ob = _case(mk, 40, nz, dict(L_DAMP_CURR=1, dt_uvdamp=360), prep_damp, nsteps=2)
    assert np.any(k3.dampu_flag > 0) or np.any(k3.dampv_flag > 0)

    # freeze clamp over the whole depth, isotherm check over 80 levels, climatology reset after a trap
    def prep_frz(k3, ob):
        T = np.asarray(k3.X[:, :, 0]).copy()
        T[::3, :] = -2.2
        T[1::3, :] = 12.0
        k3.X[:, :, 0] = T
        ob.a["T"][:, 1:nzp1 + 1] = T
        _set2(k3, ob, "ocnT_clim", 8.0 + 10.0 * np.exp(-np.arange(nzp1) / 15.0)[None, :] * np.ones((k3.npts, 1)))
        _set2(k3, ob, "sal_clim", np.asarray(k3.X[:, :, 1]) * 0.5)
        k3.U[2::3, 0:3, 0] = 40.0
        ob.a["U"][2::3, 1:4] = 40.0
    sw = dict(L_NO_FREEZE=1, L_NO_ISOTHERM=1, clim_present=1, iso_bot=min(80, nz - 2), iso_thresh=0.002)
    k3, ob = _case(mk, 42, nz, sw, prep_frz, nsteps=1)
    assert np.any(k3.freeze_flag > 0.99) and np.any(k3.reset_flag < 0) and np.any(np.abs(k3.reset_flag) == 999)
    _case(mk, 42, nz, sw, prep_frz, nsteps=3)

    # relaxation and flux corrections with depth
    def prep_rel(k3, ob):
        n = k3.npts
        _set2(k3, ob, "ocnT_clim", np.asarray(k3.X[:, :, 0]) - 0.3)
        _set2(k3, ob, "sal_clim", np.asarray(k3.X[:, :, 1]) + 0.05)
        r = np.full(n, 1.0 / (30 * 86400.0))
        k3.relax_ocnT[:] = r; ob["relax_ocnT"] = r
        k3.relax_sal[:] = 2 * r; ob["relax_sal"] = 2 * r
        _set2(k3, ob, "fcorr_withz", 5.0 * np.exp(-z / 10.0) * np.linspace(-1, 1, n)[:, None])
        _set2(k3, ob, "sfcorr_withz", 1e-7 * np.cos(z / 7.0) * np.ones((n, 1)))
    k3, ob = _case(mk, 40, nz, dict(L_RELAX_OCNT=1, L_RELAX_SAL=1, L_FCORR_WITHZ=1, L_SFCORR_WITHZ=1), prep_rel)
    assert np.any(k3.ocnTcorr[:, min(64, nz - 2):] != 0) and np.any(k3.scorr[:, min(64, nz - 2):] != 0)

    # prescribed advection, all seven modes
    def prep_adv(k3, ob):
        for c in range(k3.npts):
            k3.nmodeadv[c, 1] = 2
            k3.modeadv[c, 0, 1] = 1 + (c % 7)
            k3.modeadv[c, 1, 1] = 1 + ((c + 3) % 7)
            k3.advection[c, 0, 1] = 1e-6 * (1 + c % 5)
            k3.advection[c, 1, 1] = -5e-7
        ob["nmodeadv"][:, 1] = k3.nmodeadv[:, 1]
        ob["modeadv"][:, 1, :] = k3.modeadv[:, :, 1]
        ob["advection"][:, 1, :] = k3.advection[:, :, 1]
    _case(mk, 42, nz, dict(L_ADVECT=1), prep_adv, grid="stretched" if nz == 69 else "uniform")


@pytest.mark.parametrize("nz,sw", [(40, dict(LRI=0)), (69, dict(LRI=0)), (60, dict(LRI=0, LDD=1)), (40, dict(LRI=0, L_DAMP_CURR=1, dt_uvdamp=360))])
def test_without_the_richardson_mixing(mk, nz, sw):
    """LRI=.FALSE. (src/mckpp_physics_verticalmixing_kppmix_mod.F90:65-74): rimix is not called, the interior
    diffusivities stay at the zeros kppmix starts from (double diffusion adds to those), Rig is not formed; the
    boundary-layer scheme and the solves run on that.  Default and optional-physics builds."""
    def prep(k3, ob):
        if sw.get("LDD"):
            S = np.asarray(k3.X[:, :, 1]).copy()
            S[::2] = 0.4 - 0.8 * np.linspace(0, 1, nz + 1)[None, :]
            k3.X[:, :, 1] = S
            ob.a["S"][:, 1:nz + 2] = S
    k3, ob = _case(mk, 60, nz, sw, prep, nsteps=3, grid="stretched" if nz == 69 else "uniform")
    assert np.all(np.asarray(k3.Rig) == 0.0)          # never written
    assert np.any(np.asarray(k3.difm)[:, 1:nz] == 0.0)   # below the boundary layer


def test_lkpp_false_is_refused_with_the_reason(mk):
    kc = mk.KppConstFields(40)
    kc.LKPP = 0
    mk.mckpp_physics_lookup(kc)
    with pytest.raises(mk.MckppHipError, match="unassigned in the reference"):
        mk.MckppHip(kc)


def test_optional_physics_kernel_selection(mk):
    for nz, want in [(40, "k_column_ps<EXT>"), (60, "k_column_ps<EXT>"), (69, "k_column_ps<EXT>"), (150, "k_column_ps<EXT>")]:
        kc = mk.KppConstFields(nz)
        kc.L_DAMP_CURR = 1
        mk.mckpp_physics_lookup(kc)
        ctx = mk.MckppHip(kc)
        assert ctx.kernel_name == want, (nz, ctx.kernel_name)
        ctx.close()


def test_restart_and_ancillary_update_on_optional_physics_context(mk, tmp_path):
    """A restart file loaded into a fresh optional-physics context: stepping is refused until the
    relaxation inputs are resident (mckpp_hip_update_ancillaries), then continues bit-identically;
    a changed ancillary (SST0 / climatology, what mckpp_boundary_update rewrites between steps,
    src/mckpp_ocean_model_3D.F90:51-55) reaches the device without touching the prognostic state."""
    from oracle import orc

    ncol, nz = 96, 40
    sw = dict(L_RELAX_SST=1, L_RELAX_OCNT=1)

    def fill(k3, ob, dT):
        n = k3.npts
        r = np.full(n, 1.0 / (5 * 86400.0))
        k3.relax_sst[:] = r
        k3.relax_ocnT[:] = r / 6
        k3.SST0[:] = T0[:, 0] + dT
        k3.ocnT_clim[:, :] = T0 - 0.3 + dT
        if ob is not None:
            ob["relax_sst"] = r; ob["relax_ocnT"] = r / 6; ob["SST0"] = T0[:, 0] + dT
            ob.a["ocnT_clim"][:, 1:nz + 2] = T0 - 0.3 + dT

    oc, ob = cm.make_oracle(ncol, nz, init=False, exp_mode=1, **sw)
    kc, k3 = cm.make_hip_case(ncol, nz, land_every=7)
    for k, v in sw.items():
        setattr(kc, k, v)
    T0 = np.asarray(k3.X[:, :, 0]).copy()
    fill(k3, ob, 1.5)
    ctx = mk.mckpp_initialize_ocean_model(k3, kc)
    orc.init_ocean(oc, ob, 0)
    sf = cm.synth.forcing(ncol, "bench")
    ob["sflux"] = sf
    cm.set_forcing_3d(k3, sf)
    ctx.set_forcing(k3.sflux)
    ctx.step(1, 2)
    for nt in (1, 2):
        orc.physics_driver(oc, ob, nt)
    rst = tmp_path / "ext.restart"
    ctx.save_restart(rst)

    # (a) ancillaries change on the host between steps: only they are re-uploaded
    fill(k3, ob, 0.4)
    ctx.update_ancillaries(k3)
    ctx.step(3, 1)
    orc.physics_driver(oc, ob, 3)
    ctx.download(k3)
    ocean = np.flatnonzero(k3.run_physics)
    res = cm.compare(k3, ob, nz, cm.PROFILE_FIELDS + cm.SCALAR_FIELDS + ["fcorr", "tinc_fcorr", "ocnTcorr"], active=ocean)
    assert not {k: v for k, v in res.items() if v[2] != 0}, res

    # (b) fresh optional-physics context + restart file
    kc2, k3b = cm.make_hip_case(ncol, nz, land_every=7)
    for k, v in sw.items():
        setattr(kc2, k, v)
    ctx2 = mk.MckppHip(kc2)
    ctx2.load_restart(rst, ncol)
    with pytest.raises(mk.MckppHipError, match="update_ancillaries"):
        ctx2.step(3, 1)
    fill(k3b, None, 0.4)
    ctx2.update_ancillaries(k3b)
    cm.set_forcing_3d(k3b, sf)
    ctx2.step(3, 1)
    ctx2.download(k3b)
    for n in ("U", "X", "Us", "Xs", "hmix", "kmix", "Tref", "fcorr", "tinc_fcorr", "ocnTcorr"):
        assert np.array_equal(getattr(k3, n)[ocean], getattr(k3b, n)[ocean]), n

    # (c) a file with a corrupt column map or a truncated tail is refused and the resident state survives
    raw = bytearray(rst.read_bytes())
    hdr = 8 + 6 * 4 + 2 * 8
    bad = bytearray(raw)
    bad[hdr:hdr + 4] = (10 ** 6).to_bytes(4, "little")          # ipt[0] far outside npts
    (tmp_path / "badmap").write_bytes(bad)
    with pytest.raises(mk.MckppHipError, match="column map"):
        ctx2.load_restart(tmp_path / "badmap", ncol)
    (tmp_path / "short").write_bytes(raw[:len(raw) // 2])
    with pytest.raises(mk.MckppHipError, match="truncated"):
        ctx2.load_restart(tmp_path / "short", ncol)
    k3c = cm.make_hip_case(ncol, nz, land_every=7)[1]
    ctx2.download(k3c)
    assert np.array_equal(k3c.X[ocean], k3b.X[ocean]) and np.array_equal(k3c.hmix[ocean], k3b.hmix[ocean])
    ctx2.close()
