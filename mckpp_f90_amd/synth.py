"""Synthetic water columns and forcing.

Closed-form profiles so every consumer (HIP path, CPU oracle, Fortran host)
starts from identical bits.  Arrays are level-fastest, shape (ncol, nzp1),
level index 0 = reference level k=1.

Grid and Coriolis follow the reference's set-up formulas
(src/mckpp_initialize_geography_mod.F90:57-88); the salinity reference value
follows src/mckpp_initialize_ocean_profiles_mod.F90:104-117; the baseline
forcing constants are the reference's no-flux-file defaults
(src/mckpp_fluxes_mod.F90:41-49, 62-69).
"""
import numpy as np

SEED = 20261003
EL = 2.50e6      # latent heat of evaporation, initialize_namelist_mod.F90:103
FLSN = 334000.0  # latent heat of fusion for snow, :105-106


def uniform_grid(nz, dmax=200.0):
    """zm[1..nzp1], hm[1..nzp1], dm[0..nz] as Fortran-indexed arrays (index 0 of zm/hm unused)."""
    zm = np.zeros(nz + 2)
    hm = np.zeros(nz + 2)
    dm = np.zeros(nz + 1)
    hsum = 0.0
    for i in range(1, nz + 1):
        hm[i] = dmax / float(nz)
        zm[i] = 0.0 - (hsum + 0.5 * hm[i])
        hsum = hsum + hm[i]
        dm[i] = hsum
    hm[nz + 1] = 1.0e-10
    zm[nz + 1] = -dmax
    return zm, hm, dm


def stretched_grid(nz, dmax=1000.0, dscale=4.0):
    """The reference's l_stretchgrid option (initialize_geography_mod.F90:45-68)."""
    hm = np.zeros(nz + 2)
    zm = np.zeros(nz + 2)
    dm = np.zeros(nz + 1)
    dfac = 1.0 - np.exp(-dscale)
    sumh = 0.0
    for i in range(1, nz + 1):
        sk = -(float(i) - 0.5) / float(nz)
        hm[i] = dmax * dfac / float(nz) / dscale / (1.0 + sk * dfac)
        sumh += hm[i]
    hsum = 0.0
    for i in range(1, nz + 1):
        hm[i] = hm[i] * dmax / sumh
        zm[i] = 0.0 - (hsum + 0.5 * hm[i])
        hsum += hm[i]
        dm[i] = hsum
    hm[nz + 1] = 1.0e-10
    zm[nz + 1] = -dmax
    return zm, hm, dm


def coriolis(dlat):
    twopi = 8.0 * np.arctan(1.0)
    dlat = np.asarray(dlat, dtype=np.float64)
    small = np.abs(dlat) < 2.5
    lat_eff = np.where(small, 2.5 * np.where(dlat < 0, -1.0, 1.0), dlat)
    return 2.0 * (twopi / 86164.0) * np.sin(lat_eff * twopi / 360.0)


def forcing(ncol, mix="bench", t_seconds=None, index=None):
    """sflux(1:6,5,0) per column, shape (ncol, 6).

    mix="baseline": the reference's constant forcing on every column.
    mix="bench":    1/3 stable, 1/3 convective, 1/3 windy.
    t_seconds: if given, short-wave follows max(0, 800 sin(2 pi t / 86400)).
    """
    taux = np.full(ncol, 0.01)
    tauy = np.zeros(ncol)
    swf = np.full(ncol, 200.0)
    lhf = np.full(ncol, -150.0)
    rain = np.full(ncol, 6e-5)
    if mix == "bench":
        cls = (np.arange(ncol) if index is None else np.asarray(index)) % 3
        swf[cls == 1] = 0.0
        lhf[cls == 1] = -400.0
        taux[cls == 2] = 0.3
    elif mix != "baseline":
        raise ValueError(mix)
    if t_seconds is not None:
        diurnal = max(0.0, 800.0 * np.sin(2.0 * np.pi * t_seconds / 86400.0))
        swf = np.where(swf > 0.0, diurnal, 0.0)
    sflux = np.zeros((ncol, 6))
    sflux[:, 0] = taux
    sflux[:, 1] = tauy
    sflux[:, 2] = swf
    sflux[:, 3] = lhf            # lwf + lhf + shf - snow*flsn with lwf = shf = snow = 0
    sflux[:, 4] = 1e-10          # melting of sea ice
    sflux[:, 5] = rain + lhf / EL
    return sflux


def flux_series(ncol, nt_first, nsteps, dto, mix="bench", index=None):
    """Surface forcing records for model steps nt_first .. nt_first+nsteps-1 in the layout of
    mckpp_hip_set_flux_series, shape (nsteps, 8, ncol): taux, tauy, swf, lwf, lhf, shf, rain, snow as
    kpp_3d_fields holds them after the flux reader.  The `mix` of forcing() with the diurnal short-wave
    cycle swf = max(0, 800 sin(2 pi t / 86400)), t = (nt-1) dto; mckpp_fluxes assembles the same
    sflux(1:6) from them that forcing(t_seconds=t) returns."""
    series = np.zeros((nsteps, 8, ncol))
    sf = forcing(ncol, mix, t_seconds=0.0, index=index)        # everything but swf is constant in time
    lit = forcing(ncol, mix, index=index)[:, 2] > 0.0          # columns that see the sun at all
    series[:, 0] = sf[:, 0]                      # taux
    series[:, 1] = sf[:, 1]                      # tauy
    series[:, 4] = sf[:, 3]                      # lhf (lwf = shf = snow = 0)
    series[:, 6] = sf[:, 5] - sf[:, 3] / EL      # rain
    for r in range(nsteps):
        t_seconds = (nt_first - 1 + r) * dto
        series[r, 2] = np.where(lit, max(0.0, 800.0 * np.sin(2.0 * np.pi * t_seconds / 86400.0)), 0.0)   # swf
    return series


FLUX_NAMES = ("taux", "tauy", "swf", "lwf", "lhf", "shf", "rain", "snow")


def columns(ncol, nz, dmax=200.0, zm=None, index=None, ntotal=None):
    """Initial T, S(minus Sref), U, V on the grid and the per-column scalars.

    index/ntotal select a subset of a larger closed-form set (column i of
    ntotal), so ranks and CPU samples can take slices of one global workload."""
    if zm is None:
        zm, _, _ = uniform_grid(nz, dmax)
    nzp1 = nz + 1
    z = -zm[1:nzp1 + 1]                       # depth, positive down
    if index is None:
        i = np.arange(ncol, dtype=np.float64)
        ntotal = ncol
    else:
        i = np.asarray(index, dtype=np.float64)
        assert len(i) == ncol and ntotal is not None
    lat = -60.0 + 120.0 * i / max(ntotal - 1, 1)
    coslat = np.cos(np.deg2rad(lat))
    T = 10.0 + 18.0 * coslat[:, None] * np.exp(-np.maximum(z - 20.0, 0.0) / 80.0)[None, :]
    Sfull = np.broadcast_to(35.0 + 0.5 * z / 200.0, (ncol, nzp1)).copy()
    Sref = (Sfull[:, 0] + Sfull[:, nzp1 - 1]) / 2.0
    S = Sfull - Sref[:, None]
    U = np.broadcast_to(0.05 * np.exp(-z / 30.0), (ncol, nzp1)).copy()
    V = np.zeros((ncol, nzp1))
    return {
        "T": np.ascontiguousarray(T), "S": np.ascontiguousarray(S), "U": U, "V": V,
        "lat": lat, "f": coriolis(lat), "Sref": Sref, "SSref": Sref.copy(), "Ssurf": Sref.copy(),
        "ocdepth": np.full(ncol, -10000.0), "jerlov": np.full(ncol, 3, dtype=np.int32),
    }
