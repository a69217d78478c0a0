"""Shared helpers for the parity tests: identical synthetic inputs for the HIP
path (reference-layout Kpp3dFields over the C-ABI) and the CPU oracle."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from mckpp_f90_amd import synth  # noqa: E402
from oracle import orc  # noqa: E402


def grid_for(nz, kind="uniform"):
    if kind == "uniform":
        return synth.uniform_grid(nz, 200.0)
    if kind == "stretched":
        return synth.stretched_grid(nz, 1000.0, 4.0)
    raise ValueError(kind)


def make_oracle(ncol, nz, mix="bench", exp_mode=1, grid="uniform", dto=3600.0, nthreads=0, init=True,
                index=None, ntotal=None, **sw):
    """Oracle const + batch with synthetic columns; runs init_ocean unless init=False."""
    zm, hm, dm = grid_for(nz, grid)
    sw = {k: v for k, v in sw.items() if k != "L_ADVECT"}   # the oracle keys advection on nmodeadv alone
    # MCKPP_SOLVER_MODE sets the default solver mode of the library's handles: the checker follows it, so the whole
    # -m gpu suite can be run against either mode
    sw.setdefault("solver_mode", int(os.environ.get("MCKPP_SOLVER_MODE", "0")))
    oc = orc.Const(nz, dto=dto, exp_mode=exp_mode, zm=zm, hm=hm, dm=dm, **sw)
    col = synth.columns(ncol, nz, zm=zm, index=index, ntotal=ntotal)
    ob = orc.Batch(ncol, nz)
    for k in "UVTS":
        ob.a[k][:, 1:nz + 2] = col[k]
    ob["U_init"][:, 1:nz + 2] = col["U"]
    ob["V_init"][:, 1:nz + 2] = col["V"]
    for k in ("f", "Sref", "SSref", "Ssurf", "ocdepth", "jerlov"):
        ob[k] = col[k]
    ob["sflux"] = 1e-20      # mckpp_initialize_fluxes (src/mckpp_fluxes_mod.F90:19-32)
    if init:
        orc.init_ocean(oc, ob, ntime=0, nthreads=nthreads)
        ob["sflux"] = synth.forcing(ncol, mix, index=index)
    return oc, ob


def make_hip_case(ncol, nz, grid="uniform", dto=3600.0, land_every=0, index=None, ntotal=None):
    """KppConstFields + Kpp3dFields with the same synthetic columns (HIP side)."""
    import mckpp_f90_amd as mk

    zm, hm, dm = grid_for(nz, grid)
    kc = mk.KppConstFields(nz, dto=dto, zm=zm[1:nz + 2], hm=hm[1:nz + 2], dm=dm)
    mk.mckpp_physics_lookup(kc)
    col = synth.columns(ncol, nz, zm=zm, index=index, ntotal=ntotal)
    k3 = mk.Kpp3dFields(ncol, kc)
    k3.U[:, :, 0] = col["U"]
    k3.U[:, :, 1] = col["V"]
    k3.X[:, :, 0] = col["T"]
    k3.X[:, :, 1] = col["S"]
    k3.U_init[...] = k3.U
    for k in ("f", "Sref", "SSref", "Ssurf", "ocdepth"):
        getattr(k3, k)[:] = col[k]
    k3.jerlov[:] = col["jerlov"]
    k3.sflux[:, :, 4, 0] = 1e-20
    if land_every:
        k3.run_physics[::land_every] = 0
        k3.l_ocean[::land_every] = 0
    return kc, k3


def set_forcing_3d(k3, sflux6):
    k3.sflux[:, 0:6, 4, 0] = sflux6


# (3d field name, component index tuple, batch field, first reference index, count)
def _pairs(nz):
    nzp1 = nz + 1
    P = []
    for nm, a, l in (("U", "U", 0), ("V", "U", 1), ("T", "X", 0), ("S", "X", 1)):
        P.append((nm, a, (slice(None), slice(None), l), 1, nzp1))
    for t in (0, 1):
        for nm, a, l in (("Us", "Us", 0), ("Vs", "Us", 1), ("Ts", "Xs", 0), ("Ss", "Xs", 1)):
            P.append((f"{nm}{t}", a, (slice(None), slice(None), l, t), 1, nzp1))
    return P


def batch_to_3d_view(ob, name, lo, n):
    return ob.a[name][:, lo:lo + n]


PROFILE_FIELDS = ["U", "V", "T", "S", "Us0", "Us1", "Vs0", "Vs1", "Ts0", "Ts1", "Ss0", "Ss1"]
SCALAR_FIELDS = ["hmix", "kmix", "Tref", "uref", "vref", "Ssurf", "reset_flag"]
EXT_SCALARS = ["fcorr", "freeze_flag", "dampu_flag", "dampv_flag"]
DIAG_FIELDS = {  # batch name -> (3d array, component or None, first index, count(nz))
    "rho": ("rho", None, 0, lambda nz: nz + 2), "cp": ("cp", None, 0, lambda nz: nz + 2),
    "buoy": ("buoy", None, 1, lambda nz: nz + 1),
    "difm": ("difm", None, 0, lambda nz: nz + 2), "difs": ("difs", None, 0, lambda nz: nz + 2),
    "dift": ("dift", None, 0, lambda nz: nz + 2), "ghat": ("ghat", None, 1, lambda nz: nz),
    "wU1": ("wU", 0, 0, lambda nz: nz + 1), "wU2": ("wU", 1, 0, lambda nz: nz + 1),
    "wX1": ("wX", 0, 0, lambda nz: nz + 1), "wX2": ("wX", 1, 0, lambda nz: nz + 1),
    "wX3": ("wX", 2, 0, lambda nz: nz + 1), "wXNT1": ("wXNT", 0, 0, lambda nz: nz + 1),
    "Rig": ("Rig", None, 1, lambda nz: nz), "dbloc": ("dbloc", None, 1, lambda nz: nz),
    "Shsq": ("Shsq", None, 1, lambda nz: nz),
    "swfrac": ("swfrac", None, 1, lambda nz: nz + 1), "swdk_opt": ("swdk_opt", None, 0, lambda nz: nz + 1),
    "tinc_fcorr": ("tinc_fcorr", None, 1, lambda nz: nz + 1), "sinc_fcorr": ("sinc_fcorr", None, 1, lambda nz: nz + 1),
    "ocnTcorr": ("ocnTcorr", None, 1, lambda nz: nz + 1), "scorr": ("scorr", None, 1, lambda nz: nz + 1),
}
EXT_FIELDS = ["tinc_fcorr", "sinc_fcorr", "ocnTcorr", "scorr", "fcorr", "freeze_flag", "dampu_flag", "dampv_flag"]


def hip_field(k3, name, nz):
    """Return the HIP-side array matching oracle batch field `name`, shape (npts, count)."""
    nzp1 = nz + 1
    m = {"U": ("U", 0), "V": ("U", 1), "T": ("X", 0), "S": ("X", 1)}
    if name in m:
        a, l = m[name]
        return getattr(k3, a)[:, :, l], 1, nzp1
    ms = {"Us": ("Us", 0), "Vs": ("Us", 1), "Ts": ("Xs", 0), "Ss": ("Xs", 1)}
    if name[:2] in ms and name[2:] in ("0", "1"):
        a, l = ms[name[:2]]
        return getattr(k3, a)[:, :, l, int(name[2:])], 1, nzp1
    arr, comp, lo, cnt = DIAG_FIELDS[name]
    a = getattr(k3, arr)
    n = cnt(nz)
    if comp is not None:
        a = a[:, :, comp]
    if lo == 0:
        return a[:, 0:n], 0, n
    return a[:, 0:n], lo, n      # 1-based arrays: python index 0 <-> reference index 1


def compare(k3, ob, nz, fields, active=None):
    """Per-field (max_abs, max_rel, n_bit_mismatch) between HIP results and oracle batch."""
    out = {}
    sel = slice(None) if active is None else active
    for name in fields:
        if name in SCALAR_FIELDS or name in EXT_SCALARS or name in ("hmixd0", "hmixd1"):
            if name.startswith("hmixd"):
                h = k3.hmixd[:, int(name[-1])]
                o = ob["hmixd"][:, int(name[-1])]
            else:
                h = getattr(k3, name)
                o = ob[name]
        else:
            h, lo, n = hip_field(k3, name, nz)
            o = ob.a[name][:, lo:lo + n]
        h = np.asarray(h)[sel]
        o = np.asarray(o)[sel]
        with np.errstate(invalid="ignore"):
            d = np.abs(h - o)
        den = np.maximum(np.abs(o), 1e-300)
        # bit mismatches; +0/-0 and NaN/NaN (sign and payload of a NaN are the platform's choice) count as equal
        differ = np.ascontiguousarray(h).view(np.int64) != np.ascontiguousarray(o).view(np.int64)
        bits = int((differ & ~((h == 0) & (o == 0)) & ~(np.isnan(h) & np.isnan(o))).sum())
        out[name] = (float(d.max()) if d.size else 0.0, float((d / den).max()) if d.size else 0.0, bits)
    return out


# ---------------------------------------------------------------------------
# tolerance tables (tools/r04_tolerance.py, test_tolerance_vs_faithful_oracle)
# ---------------------------------------------------------------------------
def oracle_state(ob, nz):
    """hmix, kmix and the T/S/U/V profiles of an oracle batch as {name: array} (views)."""
    s = {k: ob.a[k][:, 1:nz + 2] for k in "TSUV"}
    s["hmix"], s["kmix"] = ob["hmix"], ob["kmix"]
    return s


def hip_state(k3, nz):
    s = {k: hip_field(k3, k, nz)[0] for k in "TSUV"}
    s["hmix"], s["kmix"] = np.asarray(k3.hmix), np.asarray(k3.kmix)
    return s


def tolerance_metrics(a, b, same_path=None, active=None):
    """SURVEY 8(d) parity gate between two runs `a` and `b` (dicts from oracle_state / hip_state): columns whose kmix
    differs now, and max / 99.9-percentile error of hmix (relative) and of T, S, U, V (relative to the profile's
    largest magnitude) - over the columns that have taken the same discrete path so far (`same_path`, default: same
    kmix now) and over the others separately."""
    sel = np.ones(len(a["hmix"]), bool) if active is None else np.asarray(active, bool)
    flipped_now = (np.asarray(a["kmix"]) != np.asarray(b["kmix"])) & sel
    same = (~flipped_now if same_path is None else np.asarray(same_path, bool) & ~flipped_now) & sel
    other = sel & ~same
    out = {"columns": int(sel.sum()), "kmix_differs_now": int(flipped_now.sum()), "off_path_columns": int(other.sum()),
           "same_path": {}, "off_path": {}}
    err = {}
    for name in "TSUV":
        scale = np.maximum(np.abs(b[name]).max(axis=1), 1e-30)
        err[name] = np.abs(a[name] - b[name]).max(axis=1) / scale
    err["hmix"] = np.abs(a["hmix"] - b["hmix"]) / np.maximum(np.abs(b["hmix"]), 1e-30)
    for name, e in err.items():
        es = e[same]
        out["same_path"][name] = {"max": float(es.max()) if es.size else 0.0,
                                  "p999": float(np.quantile(es, 0.999)) if es.size else 0.0,
                                  "columns_above_1e-10": int((es > 1e-10).sum())}
        eo = e[other]
        out["off_path"][name] = {"max": float(eo.max()) if eo.size else 0.0}
    if other.any():
        out["off_path"]["max_abs_kmix_difference"] = float(np.abs(np.asarray(a["kmix"]) - np.asarray(b["kmix"]))[other].max())
        out["off_path"]["max_abs_hmix_difference_m"] = float(np.abs(a["hmix"] - b["hmix"])[other].max())
    return out
