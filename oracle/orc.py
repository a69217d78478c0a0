"""ctypes front-end for the CPU oracle (oracle/liboracle.so) and, when built,
the compiled reference pieces (oracle/_ref/libmckpp_ref.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, bench.py's cpu_baseline leg and
__graft_entry__.smoke().  The product package never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "liboracle.so")
REFLIB = os.path.join(HERE, "_ref", "libmckpp_ref.so")

NI, NJ = 890, 48
TABLE_SHAPE = (NJ + 2, NI + 2)  # C-order view of Fortran wmt(0:891,0:49)

ST_ZERO_PIVOT, ST_LONG_ITER, ST_RETRIED, ST_FAILED, ST_DODGY = 1, 2, 4, 8, 16


def build(force=False):
    """Compile liboracle.so (and _ref when the reference is mounted)."""
    if force or not os.path.exists(LIB) or (
        os.path.getmtime(LIB) < os.path.getmtime(os.path.join(HERE, "mckpp_oracle.c"))
    ):
        subprocess.check_call(["make", "-C", HERE, "liboracle.so"], stdout=subprocess.DEVNULL)
    if os.path.isdir("/root/reference/src") and not os.path.exists(REFLIB):
        subprocess.check_call(["make", "-C", HERE, "ref"], stdout=subprocess.DEVNULL)


class OrcConst(C.Structure):
    _fields_ = [
        ("nz", C.c_int), ("itermax", C.c_int), ("hmixtolfrac", C.c_double),
        ("dto", C.c_double), ("grav", C.c_double), ("vonk", C.c_double), ("sice", C.c_double),
        ("LKPP", C.c_int), ("LRI", C.c_int), ("LDD", C.c_int), ("L_SSref", C.c_int),
        ("L_RELAX_SST", C.c_int), ("L_RELAX_CALCONLY", C.c_int), ("L_FCORR", C.c_int),
        ("L_FCORR_WITHZ", C.c_int), ("L_SFCORR", C.c_int), ("L_SFCORR_WITHZ", C.c_int),
        ("L_RELAX_SAL", C.c_int), ("L_RELAX_OCNT", C.c_int),
        ("L_NO_FREEZE", C.c_int), ("L_NO_ISOTHERM", C.c_int), ("L_DAMP_CURR", C.c_int),
        ("clim_present", C.c_int), ("iso_bot", C.c_int), ("iso_thresh", C.c_double),
        ("dt_uvdamp", C.c_int), ("exp_mode", C.c_int), ("solver_mode", C.c_int),
        ("zm", C.POINTER(C.c_double)), ("hm", C.POINTER(C.c_double)), ("dm", C.POINTER(C.c_double)),
        ("tri0", C.POINTER(C.c_double)), ("tri1", C.POINTER(C.c_double)),
        ("wmt", C.POINTER(C.c_double)), ("wst", C.POINTER(C.c_double)),
    ]


_lib = None
_ref = None


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(LIB)
        L.orc_exp_portable.restype = C.c_double
        L.orc_exp_portable.argtypes = [C.c_double]
        L.orc_cpsw.restype = C.c_double
        L.orc_cpsw.argtypes = [C.c_double] * 3
        L.orc_abk80.argtypes = [C.c_double] * 3 + [C.POINTER(C.c_double)] * 5
        L.orc_abk80_batch.argtypes = [C.c_int] + [C.POINTER(C.c_double)] * 7
        L.orc_cpsw_batch.argtypes = [C.c_int] + [C.POINTER(C.c_double)] * 4
        L.orc_z121.argtypes = [C.c_int, C.c_double, C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.orc_lookup.argtypes = [C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.orc_lookup_mode.argtypes = [C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_int]
        L.orc_conv_probe.argtypes = [C.c_int] + [C.POINTER(C.c_double)] * 6
        L.orc_conv_literals.argtypes = [C.POINTER(C.c_double)]
        L.orc_conv_unary.argtypes = [C.c_int] + [C.POINTER(C.c_double)] * 5
        L.orc_conv_swfrac.argtypes = [C.c_int, C.POINTER(C.c_double), C.c_double, C.c_int] + [C.POINTER(C.c_double)] * 2
        L.orc_conv_jerlov.argtypes = [C.c_int, C.POINTER(C.c_double)]
        L.orc_conv_binary.argtypes = [C.c_int] + [C.POINTER(C.c_double)] * 14
        L.orc_conv_casts.argtypes = [C.c_int, C.POINTER(C.c_double)] + [C.POINTER(C.c_int)] * 3 + [C.POINTER(C.c_double)]
        L.orc_wscale.argtypes = [C.POINTER(OrcConst)] + [C.c_double] * 4 + [C.POINTER(C.c_double)] * 2
        L.orc_swfrac.restype = C.c_double
        L.orc_swfrac.argtypes = [C.POINTER(OrcConst), C.c_double, C.c_double, C.c_int]
        L.orc_swdk.restype = C.c_double
        L.orc_swdk.argtypes = [C.POINTER(OrcConst), C.c_double, C.c_int]
        L.orc_tridcof.argtypes = [C.POINTER(OrcConst), C.POINTER(C.c_double), C.c_int] + [C.POINTER(C.c_double)] * 3
        L.orc_tridmat.restype = C.c_int
        L.orc_tridmat.argtypes = [C.POINTER(C.c_double)] * 5 + [C.c_int] + [C.POINTER(C.c_double)] * 2
        L.orc_tridmat_2e.restype = C.c_int
        L.orc_tridmat_2e.argtypes = [C.POINTER(C.c_double)] * 5 + [C.c_int] + [C.POINTER(C.c_double)] * 2
        L.orc_make_grid_uniform.argtypes = [C.c_int, C.c_double] + [C.POINTER(C.c_double)] * 3
        L.orc_make_tri.argtypes = [C.POINTER(OrcConst)]
        L.orc_coriolis.restype = C.c_double
        L.orc_coriolis.argtypes = [C.c_double]
        L.orc_batch_new.restype = C.c_void_p
        L.orc_batch_new.argtypes = [C.c_long, C.c_int]
        L.orc_batch_free.argtypes = [C.c_void_p]
        L.orc_batch_set.restype = C.c_int
        L.orc_batch_set.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p]
        for fn in ("orc_init_ocean", "orc_physics_driver", "orc_vmix_batch", "orc_vmix_only_batch"):
            getattr(L, fn).argtypes = [C.POINTER(OrcConst), C.c_void_p, C.c_int, C.c_int]
        L.orc_bottomtemp.argtypes = [C.POINTER(OrcConst), C.c_void_p, C.POINTER(C.c_double)]
        L.orc_fluxes.argtypes = [C.POINTER(OrcConst), C.c_void_p, C.c_int] + [C.POINTER(C.c_double)] * 8 + [C.c_int, C.c_double, C.c_double]
        _lib = L
    return _lib


PROBELIB = os.path.join(HERE, "_ref", "libconv_probe.so")
_probe = None


def conv_probe():
    """amdflang build of oracle/conv_probe.F90 (own source; needs the compiler only), or None."""
    global _probe
    if _probe is None:
        if os.path.exists("/opt/rocm/bin/amdflang") and (
                not os.path.exists(PROBELIB) or os.path.getmtime(PROBELIB) < os.path.getmtime(os.path.join(HERE, "conv_probe.F90"))):
            subprocess.check_call(["make", "-C", HERE, "probe"], stdout=subprocess.DEVNULL)
        if not os.path.exists(PROBELIB):
            return None
        P = C.CDLL(PROBELIB)
        P.conv_probe_powers.argtypes = [C.c_int] + [C.POINTER(C.c_double)] * 6
        P.conv_probe_literals.argtypes = [C.POINTER(C.c_double)]
        if hasattr(P, "conv_probe_unary"):   # (a library built from the round-4 source has only the two above)
            P.conv_probe_unary.argtypes = [C.c_int] + [C.POINTER(C.c_double)] * 5
            P.conv_probe_swfrac.argtypes = [C.c_int, C.POINTER(C.c_double)] + [C.c_double] * 4 + [C.POINTER(C.c_double)] * 2
            P.conv_probe_binary.argtypes = [C.c_int] + [C.POINTER(C.c_double)] * 14
            P.conv_probe_casts.argtypes = [C.c_int, C.POINTER(C.c_double)] + [C.POINTER(C.c_int)] * 3 + [C.POINTER(C.c_double)]
        _probe = P
    return _probe


def have_ref():
    build()
    return os.path.exists(REFLIB)


def ref():
    """Compiled reference EOS + z121 (only available where /root/reference was)."""
    global _ref
    if _ref is None:
        R = C.CDLL(REFLIB)
        R.ref_cpsw.restype = C.c_double
        R.ref_cpsw.argtypes = [C.c_double] * 3
        R.ref_abk80.argtypes = [C.c_double] * 3 + [C.POINTER(C.c_double)] * 5
        R.ref_abk80_batch.argtypes = [C.c_int] + [C.POINTER(C.c_double)] * 7
        R.ref_cpsw_batch.argtypes = [C.c_int] + [C.POINTER(C.c_double)] * 4
        R.ref_z121.argtypes = [C.c_int, C.c_double, C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        _ref = R
    return _ref


# ---------------------------------------------------------------------------
# constants
# ---------------------------------------------------------------------------
class Const:
    """Owns the numpy buffers behind an OrcConst (Fortran-indexed arrays)."""

    def __init__(self, nz, dto=3600.0, dmax=200.0, exp_mode=0, zm=None, hm=None, dm=None, half_pow_mode=0, **sw):
        L = lib()
        self.nz, self.nzp1 = nz, nz + 1
        n = nz + 4
        self.zm = np.zeros(n)
        self.hm = np.zeros(n)
        self.dm = np.zeros(n)
        self.tri0 = np.zeros(n)
        self.tri1 = np.zeros(n)
        self.wmt = np.zeros(TABLE_SHAPE)
        self.wst = np.zeros(TABLE_SHAPE)
        if zm is None:
            L.orc_make_grid_uniform(nz, dmax, _dp(self.zm), _dp(self.hm), _dp(self.dm))
        else:  # caller-supplied grid, Fortran-indexed (zm[1..nzp1], hm[1..nzp1], dm[0..nz])
            self.zm[: len(zm)] = zm
            self.hm[: len(hm)] = hm
            self.dm[: len(dm)] = dm
        c = OrcConst()
        c.nz = nz
        c.itermax = 200            # initialize_namelist_mod.F90:31
        c.hmixtolfrac = 0.1        # :32
        c.dto = dto
        c.grav, c.vonk, c.sice = 9.816, 0.4, 4.0   # :96-102
        c.LKPP, c.LRI, c.LDD, c.L_SSref = 1, 1, 0, 1  # :111-119
        c.iso_bot, c.iso_thresh, c.dt_uvdamp = 2, 0.002, 360
        c.exp_mode = exp_mode
        for k, v in sw.items():
            if not hasattr(c, k):
                raise KeyError(k)
            setattr(c, k, v)
        c.zm, c.hm, c.dm = _dp(self.zm), _dp(self.hm), _dp(self.dm)
        c.tri0, c.tri1 = _dp(self.tri0), _dp(self.tri1)
        c.wmt, c.wst = _dp(self.wmt), _dp(self.wst)
        self.c = c
        self.half_pow_mode = half_pow_mode
        L.orc_lookup_mode(c.vonk, c.wmt, c.wst, int(half_pow_mode))
        L.orc_make_tri(C.byref(c))

    @property
    def ptr(self):
        return C.byref(self.c)


# ---------------------------------------------------------------------------
# batches
# ---------------------------------------------------------------------------
LEVEL_FIELDS = [
    "U", "V", "T", "S", "Us0", "Us1", "Vs0", "Vs1", "Ts0", "Ts1", "Ss0", "Ss1", "U_init", "V_init",
    "swfrac", "swdk_opt", "rho", "cp", "buoy", "talpha", "sbeta", "difm", "difs", "dift", "ghat",
    "wU1", "wU2", "wX1", "wX2", "wX3", "wXNT1", "Rig", "dbloc", "Shsq",
    "tinc_fcorr", "sinc_fcorr", "ocnTcorr", "scorr", "fcorr_withz", "sfcorr_withz", "ocnT_clim", "sal_clim",
]
SCALAR_FIELDS = [
    "f", "Ssurf", "Sref", "SSref", "ocdepth", "hmix", "kmix", "uref", "vref", "Tref",
    "reset_flag", "dampu_flag", "dampv_flag", "freeze_flag", "fcorr",
    "relax_sst", "SST0", "fcorr_twod", "relax_sal", "relax_ocnT",
]
INT_FIELDS = ["old", "newi", "jerlov", "l_initflag", "l_ocean", "status", "npasses"]


class Batch:
    """Level-fastest batch of columns: arrays [ncol, ld], Fortran index = column index."""

    def __init__(self, ncol, nz, fields=None):
        self.ncol, self.nz, self.nzp1 = ncol, nz, nz + 1
        self.ld = nz + 4
        self.a = {}
        want = set(fields) if fields is not None else None
        for n in LEVEL_FIELDS:
            if want is None or n in want:
                self.a[n] = np.zeros((ncol, self.ld))
        for n in SCALAR_FIELDS:
            if want is None or n in want:
                self.a[n] = np.zeros(ncol)
        for n in INT_FIELDS:
            if want is None or n in want:
                self.a[n] = np.zeros(ncol, dtype=np.int32)
        self.a["nmodeadv"] = np.zeros((ncol, 2), dtype=np.int32)
        self.a["modeadv"] = np.zeros((ncol, 2, 6), dtype=np.int32)
        self.a["advection"] = np.zeros((ncol, 2, 6))
        self.a["sflux"] = np.zeros((ncol, 6))
        self.a["hmixd"] = np.zeros((ncol, 2))
        self.a["ocdepth"] = np.full(ncol, -10000.0)
        self.a["jerlov"] = np.full(ncol, 3, dtype=np.int32)
        self.a["l_ocean"] = np.ones(ncol, dtype=np.int32)
        self.a["newi"] = np.ones(ncol, dtype=np.int32)
        self._h = None

    def __getitem__(self, k):
        return self.a[k]

    def __setitem__(self, k, v):
        self.a[k][...] = v

    def copy(self):
        b = Batch.__new__(Batch)
        b.ncol, b.nz, b.nzp1, b.ld = self.ncol, self.nz, self.nzp1, self.ld
        b.a = {k: v.copy() for k, v in self.a.items()}
        b._h = None
        return b

    def handle(self):
        L = lib()
        h = L.orc_batch_new(self.ncol, self.ld)
        for k, v in self.a.items():
            assert v.flags["C_CONTIGUOUS"]
            rc = L.orc_batch_set(h, k.encode(), v.ctypes.data_as(C.c_void_p))
            assert rc == 0, k
        return h


def _run(fn, const, batch, ntime, nthreads):
    L = lib()
    h = batch.handle()
    try:
        getattr(L, fn)(const.ptr, h, int(ntime), int(nthreads))
    finally:
        L.orc_batch_free(h)


def init_ocean(const, batch, ntime=0, nthreads=0):
    _run("orc_init_ocean", const, batch, ntime, nthreads)


def physics_driver(const, batch, ntime, nthreads=0):
    _run("orc_physics_driver", const, batch, ntime, nthreads)


def vmix_batch(const, batch, ntime, nthreads=0):
    _run("orc_vmix_batch", const, batch, ntime, nthreads)


def vmix_only(const, batch, ntime, nthreads=0):
    _run("orc_vmix_only_batch", const, batch, ntime, nthreads)


def bottomtemp(const, batch, bottom_temp):
    L = lib()
    h = batch.handle()
    bt = np.ascontiguousarray(bottom_temp, dtype=np.float64)
    try:
        L.orc_bottomtemp(const.ptr, h, _dp(bt))
    finally:
        L.orc_batch_free(h)


def fluxes(const, batch, ntime, taux, tauy, swf, lwf, lhf, shf, rain, snow, l_rest=0, flsn=334000.0, el=2.5e6):
    L = lib()
    h = batch.handle()
    arrs = [np.ascontiguousarray(a, dtype=np.float64) for a in (taux, tauy, swf, lwf, lhf, shf, rain, snow)]
    try:
        L.orc_fluxes(const.ptr, h, int(ntime), *[_dp(a) for a in arrs], int(l_rest), float(flsn), float(el))
    finally:
        L.orc_batch_free(h)
