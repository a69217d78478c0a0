"""Host-side mirror of the reference's call surface over the C-ABI.

Names follow the reference so a parity test reads like the reference's own
driver (src/mckpp_ocean_model_3D.F90:23-58):

    kpp_const_fields = KppConstFields(nz=60)            # mckpp_initialize_namelist + geography
    mckpp_physics_lookup(kpp_const_fields)              # src/mckpp_physics_lookup_mod.F90:11
    kpp_3d_fields = Kpp3dFields(npts, kpp_const_fields) # mckpp_allocate_3d_fields
    ... fill U, X, f, Sref, sflux ...
    mckpp_initialize_ocean_model(kpp_3d_fields, kpp_const_fields)   # src/mckpp_initialize_ocean.F90:18
    mckpp_physics_driver(kpp_3d_fields, kpp_const_fields, ntime)    # src/mckpp_physics_driver_mod.F90:15

Arrays are numpy arrays in Fortran order with the reference's shapes
(src/mckpp_data_fields.F90:353-447), so `U[ipt, k-1, l-1]` is the reference's
`U(ipt,k,l)`; arrays with a 0 lower bound (rho, cp, difm, wU, ...) are indexed
with the reference index directly.  All compute happens in libmckpp_hip.so.
"""
import ctypes as C

import numpy as np

NI, NJ = 890, 48

F_PROFILES, F_SAVED, F_SCALARS, F_DIAG = 1, 2, 4, 8
F_RESTART = F_PROFILES | F_SAVED | F_SCALARS
F_ALL = 0xF

ST_ZERO_PIVOT, ST_LONG_ITER, ST_RETRIED, ST_FAILED, ST_DODGY = 1, 2, 4, 8, 16

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)

_SWITCHES = [
    "LKPP", "LRI", "LDD", "L_SSref", "L_RELAX_SST", "L_RELAX_CALCONLY", "L_FCORR", "L_FCORR_WITHZ",
    "L_SFCORR", "L_SFCORR_WITHZ", "L_RELAX_SAL", "L_RELAX_OCNT", "L_NO_FREEZE", "L_NO_ISOTHERM",
    "L_DAMP_CURR", "clim_present",
]


class _ConstC(C.Structure):
    _fields_ = (
        [(n, C.c_int32) for n in ("nz", "nztmax", "nsflxs", "njdt", "itermax")]
        + [(n, C.c_int32) for n in _SWITCHES]
        + [("iso_bot", C.c_int32), ("dt_uvdamp", C.c_int32), ("maxmodeadv", C.c_int32), ("L_ADVECT", C.c_int32)]
        + [(n, C.c_double) for n in ("hmixtolfrac", "dto", "grav", "vonk", "sice", "iso_thresh")]
        + [(n, _dp) for n in ("zm", "hm", "dm", "tri", "wmt", "wst")]
    )


_STATE_D = [
    "U", "X", "Us", "Xs", "U_init", "hmixd", "f", "ocdepth", "Sref", "SSref", "Ssurf", "hmix", "kmix",
    "Tref", "uref", "vref", "reset_flag", "dampu_flag", "dampv_flag", "freeze_flag", "sflux",
]
_STATE_I = ["old", "new_", "jerlov", "l_ocean", "l_initflag", "run_physics"]
_STATE_DIAG = [
    "rho", "cp", "buoy", "difm", "difs", "dift", "wU", "wX", "wXNT", "ghat", "Rig", "Shsq", "dbloc",
    "swfrac", "swdk_opt",
]
_STATE_EXT_D = ["relax_sst", "SST0", "fcorr_twod", "relax_sal", "relax_ocnT", "fcorr", "fcorr_withz", "sfcorr_withz",
                "ocnT_clim", "sal_clim", "tinc_fcorr", "sinc_fcorr", "ocnTcorr", "scorr"]
_STATE_EXT_I = ["nmodeadv", "modeadv"]


class _StateC(C.Structure):
    _fields_ = (
        [("npts", C.c_int64)]
        + [(n, _dp) for n in _STATE_D]
        + [(n, _ip) for n in _STATE_I]
        + [(n, _dp) for n in _STATE_DIAG]
        + [(n, _dp) for n in _STATE_EXT_D]
        + [(n, _ip) for n in _STATE_EXT_I]
        + [("advection", _dp)]
    )


# output fields of mckpp_hip_window_* (MCKPP_OUT_* of include/mckpp_hip.h; the XIOS field ids of
# src/mckpp_xios_io.F90:74-210)
OUT_FIELDS = ("u", "v", "T", "S_anom", "hmix", "S", "B", "wu", "wv", "wT", "wS", "wB", "wTnt", "difm", "dift", "difs",
              "rho", "cp", "scorr", "Rig", "dbloc", "Shsq", "tinc_fcorr", "fcorr_z", "sinc_fcorr", "fcorr", "taux_in",
              "tauy_in", "solar_in", "nsolar_in", "PminusE_in", "freeze_flag", "comp_flag", "dampu_flag", "dampv_flag")
OUT = {n: i for i, n in enumerate(OUT_FIELDS)}
OP_MEAN, OP_MIN, OP_MAX, OP_INSTANT = 0, 1, 2, 3


class MckppHipError(RuntimeError):
    pass


def _bind(lib):
    if getattr(lib, "_mckpp_bound", False):
        return lib
    lib.mckpp_hip_last_error.restype = C.c_char_p
    lib.mckpp_hip_device_count.restype = C.c_int
    lib.mckpp_hip_build_id.restype = C.c_char_p
    lib.mckpp_hip_init.argtypes = [C.POINTER(_ConstC), C.c_int, C.POINTER(C.c_void_p)]
    lib.mckpp_hip_finalize.argtypes = [C.c_void_p]
    lib.mckpp_host_lookup.argtypes = [C.c_double, _dp, _dp]
    lib.mckpp_host_lookup.restype = None
    lib.mckpp_host_tri.argtypes = [C.c_int32, C.c_int32, C.c_double, _dp, _dp, _dp]
    lib.mckpp_host_tri.restype = None
    lib.mckpp_hip_upload.argtypes = [C.c_void_p, C.POINTER(_StateC)]
    lib.mckpp_hip_set_forcing.argtypes = [C.c_void_p, _dp]
    lib.mckpp_hip_set_diagnostics.argtypes = [C.c_void_p, C.c_int]
    lib.mckpp_hip_last_launch_count.argtypes = [C.c_void_p]
    lib.mckpp_hip_last_launch_count.restype = C.c_int32
    lib.mckpp_hip_set_solver_mode.argtypes = [C.c_void_p, C.c_int]
    lib.mckpp_hip_get_solver_mode.argtypes = [C.c_void_p]
    lib.mckpp_hip_window_reset.argtypes = [C.c_void_p]
    lib.mckpp_hip_window_select.argtypes = [C.c_void_p, _ip, C.c_int32]
    lib.mckpp_hip_window_accumulate.argtypes = [C.c_void_p]
    lib.mckpp_hip_window_fetch.argtypes = [C.c_void_p, C.c_int, C.c_int, _dp]
    lib.mckpp_hip_save_restart.argtypes = [C.c_void_p, C.c_char_p]
    lib.mckpp_hip_load_restart.argtypes = [C.c_void_p, C.c_char_p]
    lib.mckpp_hip_update_ancillaries.argtypes = [C.c_void_p, C.POINTER(_StateC)]
    lib.mckpp_hip_fluxes.argtypes = [C.c_void_p, C.c_int] + [_dp] * 8 + [C.c_int, C.c_double, C.c_double]
    lib.mckpp_hip_bottomtemp.argtypes = [C.c_void_p, _dp]
    lib.mckpp_hip_set_flux_series.argtypes = [C.c_void_p, C.c_int, C.c_int, _dp]
    lib.mckpp_hip_run_forced.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double]
    lib.mckpp_hip_init_ocean.argtypes = [C.c_void_p, C.c_int]
    lib.mckpp_hip_step.argtypes = [C.c_void_p, C.c_int, C.c_int]
    lib.mckpp_hip_vmix_pass.argtypes = [C.c_void_p, C.c_int]
    lib.mckpp_hip_vmix_only.argtypes = [C.c_void_p, C.c_int]
    lib.mckpp_hip_synchronize.argtypes = [C.c_void_p]
    lib.mckpp_hip_download.argtypes = [C.c_void_p, C.POINTER(_StateC), C.c_uint32]
    lib.mckpp_hip_status.argtypes = [C.c_void_p, _ip, C.POINTER(C.c_int64), _ip]
    lib.mckpp_hip_last_kernel_ms.argtypes = [C.c_void_p, _dp, _ip]
    lib.mckpp_hip_kernel_residency.argtypes = [C.c_void_p, _ip, _ip, _ip, C.POINTER(C.c_int64)]
    lib.mckpp_hip_kernel_name.argtypes = [C.c_void_p]
    lib.mckpp_hip_kernel_name.restype = C.c_char_p
    lib.mckpp_hip_ncolumns.argtypes = [C.c_void_p]
    lib.mckpp_hip_ncolumns.restype = C.c_int64
    lib.mckpp_hip_eos_batch.argtypes = [C.c_void_p, C.c_int64] + [_dp] * 7
    lib.mckpp_hip_exp_batch.argtypes = [C.c_void_p, C.c_int64, _dp, _dp]
    lib.mckpp_hip_div_batch.argtypes = [C.c_void_p, C.c_int64, _dp, _dp, _dp]
    lib.mckpp_host_shard_mask.argtypes = [C.c_int64, _ip, C.c_int32, C.c_int32, _ip]
    lib.mckpp_host_shard_mask.restype = C.c_int64
    lib.mckpp_hip_multi_init.argtypes = [C.POINTER(_ConstC), C.c_int32, _ip, C.POINTER(C.c_void_p)]
    lib.mckpp_hip_multi_finalize.argtypes = [C.c_void_p]
    lib.mckpp_hip_multi_ndev.argtypes = [C.c_void_p]
    lib.mckpp_hip_multi_ctx.argtypes = [C.c_void_p, C.c_int32]
    lib.mckpp_hip_multi_ctx.restype = C.c_void_p
    lib.mckpp_hip_multi_upload.argtypes = [C.c_void_p, C.POINTER(_StateC)]
    lib.mckpp_hip_multi_set_forcing.argtypes = [C.c_void_p, _dp]
    lib.mckpp_hip_multi_set_diagnostics.argtypes = [C.c_void_p, C.c_int]
    lib.mckpp_hip_multi_set_solver_mode.argtypes = [C.c_void_p, C.c_int]
    lib.mckpp_hip_multi_init_ocean.argtypes = [C.c_void_p, C.c_int]
    lib.mckpp_hip_multi_step.argtypes = [C.c_void_p, C.c_int, C.c_int]
    lib.mckpp_hip_multi_synchronize.argtypes = [C.c_void_p]
    lib.mckpp_hip_multi_download.argtypes = [C.c_void_p, C.POINTER(_StateC), C.c_uint32]
    lib.mckpp_hip_multi_status.argtypes = [C.c_void_p, _ip, C.POINTER(C.c_int64), _ip]
    lib.mckpp_hip_multi_ncolumns.argtypes = [C.c_void_p]
    lib.mckpp_hip_multi_ncolumns.restype = C.c_int64
    lib.mckpp_hip_multi_gather.argtypes = [C.c_void_p, C.c_int32, C.c_int32, _dp]
    lib.mckpp_hip_release_host_arrays.argtypes = [C.c_void_p]
    lib.mckpp_hip_multi_release_host_arrays.argtypes = [C.c_void_p]
    lib.mckpp_hip_multi_set_flux_series.argtypes = [C.c_void_p, C.c_int, C.c_int, _dp]
    lib.mckpp_hip_multi_run_forced.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double]
    lib.mckpp_hip_multi_window_select.argtypes = [C.c_void_p, _ip, C.c_int32]
    lib.mckpp_hip_multi_window_reset.argtypes = [C.c_void_p]
    lib.mckpp_hip_multi_window_accumulate.argtypes = [C.c_void_p]
    lib.mckpp_hip_multi_window_fetch.argtypes = [C.c_void_p, C.c_int, C.c_int, _dp]
    lib.mckpp_hip_multi_save_restart.argtypes = [C.c_void_p, C.c_char_p]
    lib.mckpp_hip_multi_load_restart.argtypes = [C.c_void_p, C.c_char_p]
    lib.mckpp_hip_multi_update_ancillaries.argtypes = [C.c_void_p, C.POINTER(_StateC)]
    lib.mckpp_hip_multi_bottomtemp.argtypes = [C.c_void_p, _dp]
    lib.mckpp_hip_multi_fluxes.argtypes = [C.c_void_p, C.c_int] + [_dp] * 8 + [C.c_int, C.c_double, C.c_double]
    lib._mckpp_bound = True
    return lib


def _lib():
    from . import load_library

    return _bind(load_library())


def build_compiler():
    lib = _lib()
    lib.mckpp_hip_build_compiler.restype = C.c_char_p
    return lib.mckpp_hip_build_compiler().decode()


def build_id():
    """Identifier of the kernel sources the loaded library was built from."""
    return _lib().mckpp_hip_build_id().decode()


def _chk(rc):
    if rc != 0:
        raise MckppHipError(_lib().mckpp_hip_last_error().decode())


def _f(shape):
    return np.zeros(shape, dtype=np.float64, order="F")


class KppConstFields:
    """kpp_const_type, hot-path subset (src/mckpp_data_fields.F90:187-346), with the
    defaults of mckpp_initialize_namelist (src/mckpp_initialize_namelist_mod.F90:27-119)
    and the uniform/stretched grid of mckpp_initialize_geography
    (src/mckpp_initialize_geography_mod.F90:45-74)."""

    def __init__(self, nz, nztmax=None, dto=3600.0, dmax=200.0, zm=None, hm=None, dm=None):
        from . import synth

        self.nz, self.nzp1 = nz, nz + 1
        self.nztmax = nztmax if nztmax is not None else nz + 1
        self.nsflxs, self.njdt, self.itermax = 9, 1, 200
        self.hmixtolfrac = 0.1
        self.dto = float(dto)
        self.grav, self.vonk, self.sice = 9.816, 0.4, 4.0
        self.iso_bot, self.iso_thresh, self.dt_uvdamp = 2, 0.002, 360
        self.maxmodeadv = 6
        self.L_ADVECT = 0
        self.L_VARY_BOTTOM_TEMP = 0
        for s in _SWITCHES:
            setattr(self, s, 0)
        self.LKPP = self.LRI = self.L_SSref = 1
        if zm is None:
            z, h, d = synth.uniform_grid(nz, dmax)
            zm, hm, dm = z[1:nz + 2], h[1:nz + 2], d
        self.zm = np.ascontiguousarray(zm, dtype=np.float64)      # zm(1:nzp1)
        self.hm = np.ascontiguousarray(hm, dtype=np.float64)      # hm(1:nzp1)
        self.dm = np.ascontiguousarray(dm, dtype=np.float64)      # dm(0:nz)
        assert self.zm.shape == (nz + 1,) and self.hm.shape == (nz + 1,) and self.dm.shape == (nz + 1,)
        self.wmt = _f((NI + 2, NJ + 2))                           # wmt(0:891,0:49)
        self.wst = _f((NI + 2, NJ + 2))
        self.tri = _f((self.nztmax + 1, 2, 1))                    # tri(0:nztmax,0:1,ngrid)
        _lib().mckpp_host_tri(nz, self.nztmax, self.dto, self.zm.ctypes.data_as(_dp),
                              self.hm.ctypes.data_as(_dp), self.tri.ctypes.data_as(_dp))

    def as_c(self):
        c = _ConstC()
        for n in ("nz", "nztmax", "nsflxs", "njdt", "itermax", "iso_bot", "dt_uvdamp", "maxmodeadv", "L_ADVECT"):
            setattr(c, n, int(getattr(self, n)))
        for n in _SWITCHES:
            setattr(c, n, int(getattr(self, n)))
        for n in ("hmixtolfrac", "dto", "grav", "vonk", "sice", "iso_thresh"):
            setattr(c, n, float(getattr(self, n)))
        for n in ("zm", "hm", "dm", "tri", "wmt", "wst"):
            setattr(c, n, getattr(self, n).ctypes.data_as(_dp))
        return c


def mckpp_physics_lookup(kpp_const_fields):
    """src/mckpp_physics_lookup_mod.F90:11 - fill wmt, wst."""
    _lib().mckpp_host_lookup(kpp_const_fields.vonk, kpp_const_fields.wmt.ctypes.data_as(_dp),
                             kpp_const_fields.wst.ctypes.data_as(_dp))


class Kpp3dFields:
    """kpp_3d_type, hot-path subset, allocated like mckpp_allocate_3d_fields
    (src/mckpp_data_fields.F90:353-447)."""

    def __init__(self, npts, c):
        nz, nzp1, nzt = c.nz, c.nzp1, c.nztmax
        self.npts = npts
        self.U = _f((npts, nzp1, 2))
        self.X = _f((npts, nzp1, 2))
        self.Us = _f((npts, nzp1, 2, 2))
        self.Xs = _f((npts, nzp1, 2, 2))
        self.U_init = _f((npts, nzp1, 2))
        self.hmixd = _f((npts, 2))
        for n in ("f", "ocdepth", "Sref", "SSref", "Ssurf", "hmix", "kmix", "Tref", "uref", "vref",
                  "reset_flag", "dampu_flag", "dampv_flag", "freeze_flag"):
            setattr(self, n, _f((npts,)))
        self.ocdepth[:] = -10000.0
        self.sflux = _f((npts, c.nsflxs, 5, c.njdt + 1))
        self.old = np.zeros(npts, dtype=np.int32)
        self.new_ = np.ones(npts, dtype=np.int32)
        self.jerlov = np.full(npts, 3, dtype=np.int32)
        self.l_ocean = np.ones(npts, dtype=np.int32)
        self.l_initflag = np.zeros(npts, dtype=np.int32)
        self.run_physics = np.ones(npts, dtype=np.int32)
        self.rho = _f((npts, nzt + 2))
        self.cp = _f((npts, nzt + 2))
        self.buoy = _f((npts, nzt + 1))
        self.difm = _f((npts, nzt + 1))
        self.difs = _f((npts, nzt + 1))
        self.dift = _f((npts, nzt + 1))
        self.wU = _f((npts, nzt + 1, 3))
        self.wX = _f((npts, nzt + 1, 3))
        self.wXNT = _f((npts, nzt + 1, 2))
        self.ghat = _f((npts, nzt))
        self.Rig = _f((npts, nzp1))
        self.Shsq = _f((npts, nzp1))
        self.dbloc = _f((npts, nz))
        self.swfrac = _f((npts, nzp1))
        self.swdk_opt = _f((npts, nz + 1))
        self.bottom_temp = _f((npts,))
        for n in ("relax_sst", "SST0", "fcorr_twod", "relax_sal", "relax_ocnT", "fcorr"):
            setattr(self, n, _f((npts,)))
        for n in ("fcorr_withz", "sfcorr_withz", "ocnT_clim", "sal_clim", "tinc_fcorr", "sinc_fcorr", "ocnTcorr", "scorr"):
            setattr(self, n, _f((npts, nzp1)))
        self.nmodeadv = np.zeros((npts, 2), dtype=np.int32, order="F")
        self.modeadv = np.zeros((npts, c.maxmodeadv, 2), dtype=np.int32, order="F")
        self.advection = _f((npts, c.maxmodeadv, 2))

    def as_c(self):
        s = _StateC()
        s.npts = self.npts
        for n in _STATE_D + _STATE_DIAG + _STATE_EXT_D + ["advection"]:
            a = getattr(self, n)
            assert a.flags["F_CONTIGUOUS"] and a.dtype == np.float64, n
            setattr(s, n, a.ctypes.data_as(_dp))
        for n in _STATE_I + _STATE_EXT_I:
            a = getattr(self, n)
            assert a.dtype == np.int32 and a.flags["F_CONTIGUOUS"], n
            setattr(s, n, a.ctypes.data_as(_ip))
        return s


class MckppHip:
    """One device context (mckpp_hip_init ... mckpp_hip_finalize)."""

    def __init__(self, kpp_const_fields, device=0):
        self._h = C.c_void_p()
        self._const = kpp_const_fields
        # The library pins (hipHostRegister) every large host array it transfers from or into and keeps the
        # registration until release_host_arrays() / close(): an array must not be freed while it is registered (a new
        # array at the same address would be transferred through stale page mappings).  So every array - or object
        # holding arrays - handed to upload / download / set_forcing / gather / window_fetch stays referenced here
        # for exactly that long.
        self._held = {}
        cc = kpp_const_fields.as_c()
        _chk(_lib().mckpp_hip_init(C.byref(cc), int(device), C.byref(self._h)))

    def _hold(self, obj):
        """Keep `obj` referenced while the library may have it pinned.  Arrays are keyed by their address range (the
        library pins arrays of at least 256 KiB only: smaller ones are not kept), objects holding arrays by identity;
        a loop that hands over fresh arrays step after step is bounded: at 64 entries everything is unpinned and
        released first (a caller that wants no re-pinning reuses its arrays, or calls release_host_arrays() itself)."""
        if isinstance(obj, np.ndarray):
            if obj.nbytes < 256 * 1024:
                return
            key = (obj.ctypes.data, obj.nbytes)
        else:
            key = id(obj)
        if key not in self._held and len(self._held) >= 64:
            self.release_host_arrays()
        self._held[key] = obj

    def close(self):
        if self._h:
            _lib().mckpp_hip_finalize(self._h)
            self._h = C.c_void_p()
        self._held = {}

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def upload(self, kpp_3d_fields):
        self._hold(kpp_3d_fields)
        s = kpp_3d_fields.as_c()
        _chk(_lib().mckpp_hip_upload(self._h, C.byref(s)))
        self._npts_cache = kpp_3d_fields.npts

    def set_forcing(self, sflux):
        assert sflux.flags["F_CONTIGUOUS"]
        self._hold(sflux)
        _chk(_lib().mckpp_hip_set_forcing(self._h, sflux.ctypes.data_as(_dp)))

    def fluxes(self, ntime, taux, tauy, swf, lwf, lhf, shf, rain, snow, l_rest=0, flsn=334000.0, el=2.5e6):
        arrs = [np.ascontiguousarray(a, dtype=np.float64) for a in (taux, tauy, swf, lwf, lhf, shf, rain, snow)]
        _chk(_lib().mckpp_hip_fluxes(self._h, int(ntime), *[a.ctypes.data_as(_dp) for a in arrs], int(l_rest),
                                     float(flsn), float(el)))

    def set_flux_series(self, rec0, fields):
        """fields[nrec, 8, npts]: taux, tauy, swf, lwf, lhf, shf, rain, snow at successive flux updates."""
        f = np.ascontiguousarray(fields, dtype=np.float64)
        assert f.ndim == 3 and f.shape[1] == 8 and f.shape[2] == self._npts_cache
        self._hold(f)
        _chk(_lib().mckpp_hip_set_flux_series(self._h, int(rec0), int(f.shape[0]), f.ctypes.data_as(_dp)))

    def run_forced(self, nt_first, nsteps, ndtocn, l_rest=0, flsn=334000.0, el=2.5e6):
        _chk(_lib().mckpp_hip_run_forced(self._h, int(nt_first), int(nsteps), int(ndtocn), int(l_rest),
                                         float(flsn), float(el)))

    def bottomtemp(self, bottom_temp):
        bt = np.ascontiguousarray(bottom_temp, dtype=np.float64)
        assert bt.shape == (self._npts_cache,)
        _chk(_lib().mckpp_hip_bottomtemp(self._h, bt.ctypes.data_as(_dp)))

    def save_restart(self, path):
        _chk(_lib().mckpp_hip_save_restart(self._h, str(path).encode()))

    def load_restart(self, path, npts):
        _chk(_lib().mckpp_hip_load_restart(self._h, str(path).encode()))
        self._npts_cache = npts

    def update_ancillaries(self, kpp_3d_fields):
        """Re-upload what mckpp_boundary_update rewrites between steps (optional-physics inputs only)."""
        self._hold(kpp_3d_fields)
        sc = kpp_3d_fields.as_c()
        _chk(_lib().mckpp_hip_update_ancillaries(self._h, C.byref(sc)))

    def window_reset(self):
        _chk(_lib().mckpp_hip_window_reset(self._h))

    def window_accumulate(self):
        _chk(_lib().mckpp_hip_window_accumulate(self._h))

    def window_select(self, fields):
        """Choose the OUT_* fields window_accumulate reduces (resets the window)."""
        f = np.ascontiguousarray(fields, dtype=np.int32)
        _chk(_lib().mckpp_hip_window_select(self._h, f.ctypes.data_as(_ip), len(f)))

    def window_fetch(self, field, op, out):
        assert out.flags["F_CONTIGUOUS"] and out.dtype == np.float64
        self._hold(out)
        _chk(_lib().mckpp_hip_window_fetch(self._h, int(field), int(op), out.ctypes.data_as(_dp)))
        return out

    def set_diagnostics(self, on):
        _chk(_lib().mckpp_hip_set_diagnostics(self._h, int(on)))

    def set_solver_mode(self, mode):
        """0: tridmat's order of operations (default); 1: two-ended elimination (mckpp_hip_set_solver_mode)."""
        _chk(_lib().mckpp_hip_set_solver_mode(self._h, int(mode)))

    @property
    def solver_mode(self):
        return int(_lib().mckpp_hip_get_solver_mode(self._h))

    def release_host_arrays(self):
        """Un-pin the caller's arrays this context registered (before they are freed while the context lives);
        the references this object holds on them go with the registrations."""
        _chk(_lib().mckpp_hip_release_host_arrays(self._h))
        self._held = {}

    def init_ocean(self, ntime=0):
        _chk(_lib().mckpp_hip_init_ocean(self._h, int(ntime)))

    def step(self, ntime, nsteps=1):
        _chk(_lib().mckpp_hip_step(self._h, int(ntime), int(nsteps)))

    def vmix_pass(self, ntime):
        _chk(_lib().mckpp_hip_vmix_pass(self._h, int(ntime)))

    def vmix_only(self, ntime):
        _chk(_lib().mckpp_hip_vmix_only(self._h, int(ntime)))

    def synchronize(self):
        _chk(_lib().mckpp_hip_synchronize(self._h))

    def download(self, kpp_3d_fields, mask=F_ALL):
        self._hold(kpp_3d_fields)
        s = kpp_3d_fields.as_c()
        _chk(_lib().mckpp_hip_download(self._h, C.byref(s), int(mask)))

    def status(self):
        n = self._npts()
        st = np.zeros(n, dtype=np.int32)
        npass = np.zeros(n, dtype=np.int32)
        nf = C.c_int64(0)
        _chk(_lib().mckpp_hip_status(self._h, st.ctypes.data_as(_ip), C.byref(nf), npass.ctypes.data_as(_ip)))
        return st, int(nf.value), npass

    def _npts(self):
        return self._npts_cache

    def last_kernel_ms(self):
        ms = C.c_double(0)
        nl = C.c_int32(0)
        _chk(_lib().mckpp_hip_last_kernel_ms(self._h, C.byref(ms), C.byref(nl)))
        return ms.value, nl.value

    def last_launch_count(self):
        """kernel launches of the last step/init/vmix_pass call (step(nt, n > 1) is one launch for all n steps)"""
        return int(_lib().mckpp_hip_last_launch_count(self._h))

    def kernel_residency(self):
        """(blocks per CU asked for, blocks per CU that fit, threads per block, LDS bytes per block)"""
        b, m, t = C.c_int32(0), C.c_int32(0), C.c_int32(0)
        l = C.c_int64(0)
        _chk(_lib().mckpp_hip_kernel_residency(self._h, C.byref(b), C.byref(m), C.byref(t), C.byref(l)))
        return b.value, m.value, t.value, l.value

    @property
    def kernel_name(self):
        return _lib().mckpp_hip_kernel_name(self._h).decode()

    @property
    def ncolumns(self):
        return int(_lib().mckpp_hip_ncolumns(self._h))

    def eos_batch(self, s, t, p):
        n = len(s)
        out = [np.zeros(n) for _ in range(4)]
        _chk(_lib().mckpp_hip_eos_batch(self._h, n, *[np.ascontiguousarray(a, dtype=np.float64).ctypes.data_as(_dp) for a in (s, t, p)],
                                        *[o.ctypes.data_as(_dp) for o in out]))
        return out

    def div_batch(self, num, den):
        num = np.ascontiguousarray(num, dtype=np.float64)
        den = np.ascontiguousarray(den, dtype=np.float64)
        q = np.zeros((4, len(num)))
        _chk(_lib().mckpp_hip_div_batch(self._h, len(num), num.ctypes.data_as(_dp), den.ctypes.data_as(_dp), q.ctypes.data_as(_dp)))
        return q

    def exp_batch(self, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        y = np.zeros_like(x)
        _chk(_lib().mckpp_hip_exp_batch(self._h, len(x), x.ctypes.data_as(_dp), y.ctypes.data_as(_dp)))
        return y


class MckppHipMulti:
    """Several GPUs behind one handle (mckpp_hip_multi_*): columns dealt round-robin to the devices."""

    def __init__(self, kpp_const_fields, devices):
        self._h = C.c_void_p()
        self._c = kpp_const_fields
        cc = kpp_const_fields.as_c()
        dev = np.ascontiguousarray(devices, dtype=np.int32)
        _chk(_lib().mckpp_hip_multi_init(C.byref(cc), len(dev), dev.ctypes.data_as(_ip), C.byref(self._h)))
        self._npts = 0
        self._held = {}   # host arrays the shards have pinned: referenced until close() / release_host_arrays() (see MckppHip)

    def _hold(self, obj):
        """Keep `obj` referenced while the library may have it pinned.  Arrays are keyed by their address range (the
        library pins arrays of at least 256 KiB only: smaller ones are not kept), objects holding arrays by identity;
        a loop that hands over fresh arrays step after step is bounded: at 64 entries everything is unpinned and
        released first (a caller that wants no re-pinning reuses its arrays, or calls release_host_arrays() itself)."""
        if isinstance(obj, np.ndarray):
            if obj.nbytes < 256 * 1024:
                return
            key = (obj.ctypes.data, obj.nbytes)
        else:
            key = id(obj)
        if key not in self._held and len(self._held) >= 64:
            self.release_host_arrays()
        self._held[key] = obj

    def close(self):
        if self._h:
            _lib().mckpp_hip_multi_finalize(self._h)
            self._h = C.c_void_p()
        self._held = {}

    def upload(self, k3):
        self._hold(k3)
        sc = k3.as_c()
        _chk(_lib().mckpp_hip_multi_upload(self._h, C.byref(sc)))
        self._npts = k3.npts

    def set_forcing(self, sflux):
        self._hold(sflux)
        _chk(_lib().mckpp_hip_multi_set_forcing(self._h, sflux.ctypes.data_as(_dp)))

    def init_ocean(self, ntime=0):
        _chk(_lib().mckpp_hip_multi_init_ocean(self._h, int(ntime)))

    def step(self, ntime, nsteps=1):
        _chk(_lib().mckpp_hip_multi_step(self._h, int(ntime), int(nsteps)))

    def synchronize(self):
        _chk(_lib().mckpp_hip_multi_synchronize(self._h))

    def download(self, k3, mask=F_ALL):
        self._hold(k3)
        sc = k3.as_c()
        _chk(_lib().mckpp_hip_multi_download(self._h, C.byref(sc), mask))

    def status(self):
        st = np.zeros(self._npts, dtype=np.int32)
        npass = np.zeros(self._npts, dtype=np.int32)
        nf = C.c_int64()
        _chk(_lib().mckpp_hip_multi_status(self._h, st.ctypes.data_as(_ip), C.byref(nf), npass.ctypes.data_as(_ip)))
        return st, nf.value, npass

    @property
    def ncolumns(self):
        return _lib().mckpp_hip_multi_ncolumns(self._h)

    def gather(self, field, root, out):
        """field 0 U, 1 V, 2 T, 3 S -> out(npts, nzp1) Fortran order; 4 hmix -> out(npts)."""
        assert out.flags["F_CONTIGUOUS"] and out.dtype == np.float64
        self._hold(out)
        _chk(_lib().mckpp_hip_multi_gather(self._h, int(field), int(root), out.ctypes.data_as(_dp)))

    def set_diagnostics(self, on):
        _chk(_lib().mckpp_hip_multi_set_diagnostics(self._h, int(on)))

    def set_solver_mode(self, mode):
        _chk(_lib().mckpp_hip_multi_set_solver_mode(self._h, int(mode)))

    def set_flux_series(self, rec0, fields):
        """fields[nrec][8][npts] (taux,tauy,swf,lwf,lhf,shf,rain,snow); record 0 is flux update rec0."""
        f = np.ascontiguousarray(fields, dtype=np.float64)
        assert f.ndim == 3 and f.shape[1] == 8 and f.shape[2] == self._npts
        self._hold(f)
        _chk(_lib().mckpp_hip_multi_set_flux_series(self._h, int(rec0), f.shape[0], f.ctypes.data_as(_dp)))

    def run_forced(self, nt_first, nsteps, ndtocn, l_rest=0, flsn=334000.0, el=2.5e6):
        _chk(_lib().mckpp_hip_multi_run_forced(self._h, int(nt_first), int(nsteps), int(ndtocn), int(l_rest),
                                               float(flsn), float(el)))

    def window_select(self, fields):
        f = np.ascontiguousarray(fields, dtype=np.int32)
        _chk(_lib().mckpp_hip_multi_window_select(self._h, f.ctypes.data_as(_ip), len(f)))

    def window_reset(self):
        _chk(_lib().mckpp_hip_multi_window_reset(self._h))

    def window_accumulate(self):
        _chk(_lib().mckpp_hip_multi_window_accumulate(self._h))

    def window_fetch(self, field, op, out):
        assert out.flags["F_CONTIGUOUS"] and out.dtype == np.float64
        self._hold(out)
        _chk(_lib().mckpp_hip_multi_window_fetch(self._h, int(field), int(op), out.ctypes.data_as(_dp)))
        return out

    def save_restart(self, path):
        _chk(_lib().mckpp_hip_multi_save_restart(self._h, str(path).encode()))

    def load_restart(self, path):
        _chk(_lib().mckpp_hip_multi_load_restart(self._h, str(path).encode()))

    def release_host_arrays(self):
        _chk(_lib().mckpp_hip_multi_release_host_arrays(self._h))
        self._held = {}


def host_shard_mask(run_physics, ndev, dev):
    """run_physics mask of shard dev of ndev (mckpp_host_shard_mask; host only)."""
    rp = np.ascontiguousarray(run_physics, dtype=np.int32)
    out = np.zeros(len(rp), dtype=np.int32)
    n = _lib().mckpp_host_shard_mask(len(rp), rp.ctypes.data_as(_ip), int(ndev), int(dev), out.ctypes.data_as(_ip))
    if n < 0:
        raise MckppHipError("mckpp_host_shard_mask: bad arguments")
    return out, int(n)


# ---------------------------------------------------------------------------
# The reference's call surface.  The reference operates on module globals; the
# Python mirror passes them explicitly and keeps the device context on the
# const-fields object (one context per kpp_const_fields, created on first use).
# ---------------------------------------------------------------------------
def _ctx(kpp_3d_fields, kpp_const_fields, device=0):
    ctx = getattr(kpp_const_fields, "_hip_ctx", None)
    if ctx is None:
        ctx = MckppHip(kpp_const_fields, device)
        kpp_const_fields._hip_ctx = ctx
        ctx._resident = None
    if ctx._resident is not kpp_3d_fields:
        ctx.upload(kpp_3d_fields)
        ctx._resident = kpp_3d_fields
    return ctx


def mckpp_initialize_ocean_model(kpp_3d_fields, kpp_const_fields, ntime=0, device=0, download=True):
    """src/mckpp_initialize_ocean.F90:18 (tri() is already set by KppConstFields)."""
    ctx = _ctx(kpp_3d_fields, kpp_const_fields, device)
    ctx.init_ocean(ntime)
    if download:
        ctx.download(kpp_3d_fields, F_ALL)
    return ctx


def mckpp_physics_driver(kpp_3d_fields, kpp_const_fields, ntime, device=0, download=True, new_forcing=True):
    """src/mckpp_physics_driver_mod.F90:15 - one ocnstep + check_profile per run_physics column."""
    ctx = _ctx(kpp_3d_fields, kpp_const_fields, device)
    if new_forcing:
        ctx.set_forcing(kpp_3d_fields.sflux)
    ctx.step(ntime, 1)
    if kpp_const_fields.L_VARY_BOTTOM_TEMP:     # src/mckpp_physics_driver_mod.F90:67-71
        ctx.bottomtemp(kpp_3d_fields.bottom_temp)
    if download:
        ctx.download(kpp_3d_fields, F_ALL)
    return ctx
