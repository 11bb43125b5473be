"""
oracle.py — CPU (numpy + C) restatement of Breeze.jl's anelastic SSP-RK3 step.

TEST INFRASTRUCTURE ONLY.  Imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg; never by the product package (breeze.jl_amd/).

PARITY STATUS: "parity unpinned" for the WENO reconstruction arithmetic and the
Oceananigans pieces (see breeze_oracle.c header).  Pinned parts: thermodynamic
constants / reference columns / Poisson solve / projection / conservation via the
reference's own known-answer tests restated in tests/; and, with REFERENCE-GENERATED
numbers (the Field summaries its model-level jldoctests print with six digits, kept in
tests/golden/reference_doctests.json): the default ReferenceState pressure column and
the theta -> T diagnosis (StaticEnergy doctest), the vapour-only virtual potential
temperature, the saturation-adjustment state + Clausius-Clapeyron pressure
(RelativeHumidity doctest, 128 levels) and the dewpoint inversion
(tests/test_golden_reference.py; the device model repeats two of them on the GPU).

Reference call stack restated here (file:line relative to /root/reference):
  time_step!            src/TimeSteppers/ssp_runge_kutta_3.jl:209-278
  ssp_rk3_substep!      src/TimeSteppers/ssp_runge_kutta_3.jl:114-173
  compute_pressure_correction! / make_pressure_correction!
                        src/AnelasticEquations/anelastic_time_stepping.jl:26-78
  solve_for_anelastic_pressure!
                        src/AnelasticEquations/anelastic_pressure_solver.jl:84-105
  update_state!         src/AtmosphereModels/update_atmosphere_model_state.jl:41-68
  compute_tendencies!   src/AtmosphereModels/update_atmosphere_model_state.jl:294-387
  set!                  src/AtmosphereModels/set_atmosphere_model.jl:198-362
  ReferenceState        src/Thermodynamics/reference_states.jl:402-445
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libbreeze_oracle.so")

PERIODIC, BOUNDED, FLAT, SLAB = 0, 1, 2, 3     # SLAB: periodic stencils, halos filled by the caller (y-slab tests)
_TOPO = {"Periodic": PERIODIC, "Bounded": BOUNDED, "Flat": FLAT, "Slab": SLAB,
         PERIODIC: PERIODIC, BOUNDED: BOUNDED, FLAT: FLAT, SLAB: SLAB}


def build(force=False):
    """Compile the C oracle with gcc (recipe: oracle/Makefile)."""
    srcs = [os.path.join(_HERE, f) for f in ("breeze_oracle.c", "breeze_oracle_compressible.inc.c")]
    c2 = _LIB_PATH.replace("libbreeze_oracle.so", "libbreeze_oracle_centered2.so")
    stale = lambda p: not os.path.exists(p) or os.path.getmtime(p) < max(map(os.path.getmtime, srcs))
    if force or stale(_LIB_PATH) or stale(c2):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB_PATH


_dp = C.POINTER(C.c_double)


class _OGGrid(C.Structure):
    _fields_ = [("Nx", C.c_int), ("Ny", C.c_int), ("Nz", C.c_int),
                ("Hx", C.c_int), ("Hy", C.c_int), ("Hz", C.c_int),
                ("tx", C.c_int), ("ty", C.c_int), ("tz", C.c_int),
                ("dx", C.c_double), ("dy", C.c_double),
                ("dzc", _dp), ("dzf", _dp),
                ("rho_r", _dp), ("p_r", _dp), ("T_r", _dp),
                ("g", C.c_double), ("Rd", C.c_double), ("Rv", C.c_double),
                ("cpd", C.c_double), ("cpv", C.c_double), ("p_st", C.c_double)]


class _OGSA(C.Structure):
    _fields_ = [("Ll", C.c_double), ("cl", C.c_double), ("T_energy", C.c_double), ("Ttr", C.c_double),
                ("ptr", C.c_double), ("abstol", C.c_double), ("maxiter", C.c_int)]


_lib = None


_libs = {}


def lib(advection="WENO5"):
    """The C oracle for advection = WENO(order = 5) (default) or "Centered2" = Centered(order = 2), the reference constructor's
    default scheme (the same sources compiled with -DOG_CENTERED2: every reconstruction is the 2-point symmetric mean)."""
    global _lib
    if advection not in ("WENO5", "Centered2"):
        raise ValueError(advection)
    if advection not in _libs:
        build()
        path = _LIB_PATH if advection == "WENO5" else _LIB_PATH.replace("libbreeze_oracle.so", "libbreeze_oracle_centered2.so")
        L = C.CDLL(path)
        L.og_weno5.restype = C.c_double
        L.og_weno5.argtypes = [C.c_double] * 5
        L.og_weno3.restype = C.c_double
        L.og_weno3.argtypes = [C.c_double] * 3
        L.og_buffer_at.restype = C.c_int
        L.og_buffer_at.argtypes = [C.c_int] * 4
        _libs[advection] = L
        if advection == "WENO5":
            _lib = L
    return _libs[advection]


def _p(a):
    assert a.dtype == np.float64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(_dp)


# ---------------------------------------------------------------------------
# Thermodynamic constants (src/Thermodynamics/thermodynamics_constants.jl:182-212)
# ---------------------------------------------------------------------------
class Constants:
    def __init__(self, molar_gas_constant=8.314462618, gravitational_acceleration=9.81,
                 dry_air_molar_mass=0.02897, dry_air_heat_capacity=1005.0,
                 vapor_molar_mass=0.018015, vapor_heat_capacity=1850.0):
        self.R = molar_gas_constant
        self.g = gravitational_acceleration
        self.Md, self.cpd = dry_air_molar_mass, float(dry_air_heat_capacity)
        self.Mv, self.cpv = vapor_molar_mass, float(vapor_heat_capacity)
        self.Rd = self.R / self.Md     # dry_air_gas_constant  (:215)
        self.Rv = self.R / self.Mv     # vapor_gas_constant    (:214)


# Closed-form adiabatic hydrostatic reference profiles (reference_states.jl:84-123,326-330)
def surface_density(p0, theta0, pst, c):
    Pi0 = (p0 / pst) ** (c.Rd / c.cpd)
    T0 = Pi0 * theta0
    return p0 / (c.Rd * T0)


def adiabatic_hydrostatic_pressure(z, p0, theta0, pst, c):
    T0 = theta0 * (p0 / pst) ** (c.Rd / c.cpd)
    return p0 * (1 - c.g * z / (c.cpd * T0)) ** (c.cpd / c.Rd)


def adiabatic_hydrostatic_density(z, p0, theta0, pst, c):
    pr = adiabatic_hydrostatic_pressure(z, p0, theta0, pst, c)
    rho0 = surface_density(p0, theta0, pst, c)
    return rho0 * (pr / p0) ** (1 - c.Rd / c.cpd)


def hydrostatic_temperature(z, p0, theta0, pst, c):
    kappa = c.Rd / c.cpd
    p = adiabatic_hydrostatic_pressure(z, p0, theta0, pst, c)
    return theta0 * (p / pst) ** kappa


# ---------------------------------------------------------------------------
# Grid (Oceananigans RectilinearGrid, regular in x,y; regular or stretched z)
# ---------------------------------------------------------------------------
class Grid:
    def __init__(self, size, x=None, y=None, z=None, topology=("Periodic", "Periodic", "Bounded"),
                 halo=None):
        topo = tuple(_TOPO[t] for t in topology)
        self.topo = topo
        size = (size,) if np.isscalar(size) else tuple(size)
        nonflat = [d for d in range(3) if topo[d] != FLAT]
        assert len(size) == len(nonflat), "size must list the non-Flat dimensions"
        N = [1, 1, 1]
        for n, d in zip(size, nonflat):
            N[d] = int(n)
        if halo is None:
            halo = (3,) * len(nonflat)
        halo = (halo,) * len(nonflat) if np.isscalar(halo) else tuple(halo)
        H = [0, 0, 0]
        for h, d in zip(halo, nonflat):
            H[d] = int(h)
        self.Nx, self.Ny, self.Nz = N
        self.Hx, self.Hy, self.Hz = H
        assert topo[2] == BOUNDED, "oracle supports Bounded z only"
        assert topo[0] in (PERIODIC, FLAT, BOUNDED) and topo[1] in (PERIODIC, FLAT, SLAB, BOUNDED)      # walls: (Periodic, Bounded, Bounded), (Bounded, Flat, Bounded)

        def regular(ext, n, flat):
            if flat:
                return 0.0, 1.0, np.zeros(1), np.zeros(1)
            a, b = ext
            d = (b - a) / n
            return a, d, a + d * np.arange(n), a + d * (np.arange(n) + 0.5)

        self.x0, self.dx, self.xf, self.xc = regular(x, self.Nx, topo[0] == FLAT)
        self.y0, self.dy, self.yf, self.yc = regular(y, self.Ny, topo[1] == FLAT)
        Nz, Hz = self.Nz, self.Hz
        if isinstance(z, (tuple, list)) and len(z) == 2:
            zf = z[0] + (z[1] - z[0]) / Nz * np.arange(Nz + 1)
            zf[-1] = z[1]
        else:
            zf = np.asarray(z, dtype=np.float64)
            assert zf.shape == (Nz + 1,)
        self.zf = zf
        # halo extension: mirror the first/last spacing outward
        zf_ext = np.empty(Nz + 1 + 2 * Hz)
        zf_ext[Hz:Hz + Nz + 1] = zf
        for h in range(1, Hz + 1):
            zf_ext[Hz - h] = zf_ext[Hz - h + 1] - (zf[1] - zf[0])
            zf_ext[Hz + Nz + h] = zf_ext[Hz + Nz + h - 1] + (zf[-1] - zf[-2])
        zc_ext = 0.5 * (zf_ext[:-1] + zf_ext[1:])           # length Nz+2Hz
        self.zc = zc_ext[Hz:Hz + Nz].copy()
        self.dzc = np.ascontiguousarray(zf_ext[1:] - zf_ext[:-1])     # entry k+Hz
        dzf = np.zeros(Nz + 1 + 2 * Hz)
        dzf[1:Nz + 2 * Hz] = zc_ext[1:] - zc_ext[:-1]                 # dzf[k+Hz] = zc[k]-zc[k-1]
        dzf[0] = dzf[1]
        dzf[-1] = dzf[-2]
        self.dzf = np.ascontiguousarray(dzf)
        self.regular_z = isinstance(z, (tuple, list)) and len(z) == 2
        if self.regular_z:          # Oceananigans regular grids carry one number dz = Lz/Nz
            dz = (z[1] - z[0]) / Nz
            self.dzc[:] = dz
            self.dzf[:] = dz
            self.zc = z[0] + dz * (np.arange(Nz) + 0.5)
        self.Sx, self.Sy = self.Nx + 2 * self.Hx, self.Ny + 2 * self.Hy
        self.Szc, self.Szf = Nz + 2 * Hz, Nz + 1 + 2 * Hz

    def center_field(self):
        return np.zeros((self.Szc, self.Sy, self.Sx))

    def zface_field(self):
        return np.zeros((self.Szf, self.Sy, self.Sx))

    def interior(self, f, zface=False):
        nz = self.Nz + (1 if zface else 0)
        return f[self.Hz:self.Hz + nz, self.Hy:self.Hy + self.Ny, self.Hx:self.Hx + self.Nx]

    def nodes(self, loc):
        """Broadcastable (z, y, x) node arrays for location loc in {'ccc','fcc','cfc','ccf'}."""
        x = self.xf if loc[0] == "f" else self.xc
        y = self.yf if loc[1] == "f" else self.yc
        z = self.zf if loc[2] == "f" else self.zc
        return x[None, None, :], y[None, :, None], z[:, None, None]


class ReferenceState:
    """ReferenceState(grid; surface_pressure, potential_temperature, standard_pressure)
    — reference_states.jl:402-445 (constant potential temperature, dry)."""

    def __init__(self, grid, constants, surface_pressure=101325.0, potential_temperature=288.0,
                 standard_pressure=1e5):
        g, c = grid, constants
        self.p0, self.theta0, self.pst = float(surface_pressure), float(potential_temperature), float(standard_pressure)
        Nz, Hz = g.Nz, g.Hz
        self.rho0 = surface_density(self.p0, self.theta0, self.pst, c)
        self.density = np.zeros(g.Szc)
        self.pressure = np.zeros(g.Szc)
        self.temperature = np.zeros(g.Szc)
        zc = g.zc
        self.density[Hz:Hz + Nz] = adiabatic_hydrostatic_density(zc, self.p0, self.theta0, self.pst, c)
        self.pressure[Hz:Hz + Nz] = adiabatic_hydrostatic_pressure(zc, self.p0, self.theta0, self.pst, c)
        self.temperature[Hz:Hz + Nz] = hydrostatic_temperature(zc, self.p0, self.theta0, self.pst, c)
        self._fill(g)

    def _fill(self, g):
        """bottom ValueBoundaryCondition (rho0 / p0), top + T: zero-gradient; first halo cell only
        (reference_states.jl:416-438)."""
        Nz, Hz = g.Nz, g.Hz
        if Hz > 0:
            self.density[Hz - 1] = 2 * self.rho0 - self.density[Hz]
            self.pressure[Hz - 1] = 2 * self.p0 - self.pressure[Hz]
            self.temperature[Hz - 1] = self.temperature[Hz]
            self.density[Hz + Nz] = self.density[Hz + Nz - 1]
            self.pressure[Hz + Nz] = self.pressure[Hz + Nz - 1]
            self.temperature[Hz + Nz] = self.temperature[Hz + Nz - 1]


def poisson_eigenvalues(N, delta, topo):
    """Oceananigans poisson_eigenvalues (recalled): Periodic (2 sin(pi (i-1)/N)/Delta)^2, Bounded (2 sin(pi (i-1)/(2N))/Delta)^2
    (the cosine modes of the staggered Neumann problem), Flat 0."""
    if topo == FLAT:
        return np.zeros(1)
    i = np.arange(N)
    if topo == BOUNDED:
        return (2 * np.sin(i * np.pi / (2 * N)) / delta) ** 2
    assert topo == PERIODIC
    return (2 * np.sin(i * np.pi / N) / delta) ** 2


class OracleModel:
    """AtmosphereModel(grid; advection=WENO(order=5), dynamics=AnelasticDynamics(ReferenceState),
    formulation=:LiquidIcePotentialTemperature, microphysics=nothing, closure=nothing) on the CPU."""

    PROGNOSTIC = ("ru", "rv", "rw", "rtheta", "rq")

    def __init__(self, grid, constants=None, surface_pressure=101325.0, potential_temperature=288.0,
                 standard_pressure=1e5, reference_density=None, initialize=True,
                 formulation="LiquidIcePotentialTemperature", microphysics=None, sa_abstol=1e-4, sa_maxiter=20,
                 forcings=None, closure=None, tracers=0, advection="WENO5", scalar_advection=None, weno_ft2=0):
        # formulation "StaticEnergy": self.theta holds e, self.rtheta holds rho*e
        # (src/StaticEnergyFormulations/static_energy_formulation.jl:18-21)
        assert formulation in ("LiquidIcePotentialTemperature", "StaticEnergy")
        self.formulation = formulation
        # microphysics "SaturationAdjustment": SaturationAdjustment(equilibrium = WarmPhaseEquilibrium()); self.q / self.rq
        # then hold the equilibrium (total) moisture q^e / rho q^e, self.qv / self.ql the diagnosed vapour and liquid
        assert microphysics in (None, "SaturationAdjustment", "Kessler")
        assert not (microphysics and formulation == "StaticEnergy"), "oracle: saturation adjustment with theta only"
        self.microphysics = microphysics
        self.forcings = forcings           # oracle.forcings.ColumnForcings (BOMEX forcing stack) or None
        self.closure = closure             # oracle.closure.SmagorinskyLilly or None
        assert not (closure and (formulation == "StaticEnergy" or microphysics == "Kessler"))
        assert not (forcings and (formulation == "StaticEnergy" or microphysics == "Kessler"))
        self.grid = g = grid
        self.constants = c = constants or Constants()
        self.ref = ReferenceState(g, c, surface_pressure, potential_temperature, standard_pressure)
        self.ref._fill(g)
        if reference_density is not None:      # test hook: set!(reference_state.density, f(z))
            self.ref.density[g.Hz:g.Hz + g.Nz] = reference_density(g.zc)
            self.ref._fill(g)
        # advection: "WENO5" (default), "WENO7", "WENO9" (the order is a process-wide switch of the C library, set before every
        # tendency evaluation; halos must be at least (order + 1) / 2 wide) or "Centered2" (a separate build of the library)
        # scalar_advection: the scalars' order where it differs from the momentum scheme (AtmosphereModel(grid; momentum_advection,
        # scalar_advection), atmosphere_model.jl:80-82,148-158; examples/tropical_cyclone_world.jl:167-169)
        # weno_ft2: 0 (default) | 1 | 2 — the FT2 hypothesis of SURVEY Appendix D.1 (breeze_oracle.c: og_set_weno_ft2)
        assert weno_ft2 in (0, 1, 2)
        self.weno_ft2 = weno_ft2
        self.weno_order = {"WENO5": 5, "WENO7": 7, "WENO9": 9}.get(advection, 5)
        self.scalar_order = {"WENO5": 5, "WENO7": 7, "WENO9": 9}[scalar_advection] if scalar_advection else self.weno_order
        assert scalar_advection is None or advection.startswith("WENO")
        if advection.startswith("WENO"):
            need = (max(self.weno_order, self.scalar_order) + 1) // 2
            assert min(g.Hx if g.Nx > 1 else need, g.Hy if g.Ny > 1 else need, g.Hz) >= need, f"{advection} needs halos >= {need}"
        self.lib = lib("WENO5" if advection.startswith("WENO") else advection)
        self._mk_cgrid()
        self.qv, self.ql = g.center_field(), g.center_field()
        # DCMIP2016KesslerMicrophysics: prognostic rho q^cl, rho q^r; diagnostic q^cl, q^r, W^r, precipitation_rate; self.q = q^v
        # (dcmip2016_kessler.jl:216-290); ql holds q^cl + q^r for the buoyancy
        if microphysics == "Kessler":
            from .kessler import KesslerParameters, TetensConstants
            self.PROGNOSTIC = OracleModel.PROGNOSTIC + ("rqcl", "rqr")
            self.rqcl, self.rqr, self.qcl, self.qr, self.W = (g.center_field() for _ in range(5))
            self.precipitation_rate = np.zeros((g.Ny, g.Nx))
            self.kessler = KesslerParameters()
            c0 = self.constants
            self.tetens = TetensConstants(molar_gas_constant=c0.R, dry_air_molar_mass=c0.Md, vapor_molar_mass=c0.Mv,
                                          dry_air_heat_capacity=c0.cpd, vapor_heat_capacity=c0.cpv)
        from .thermo import ThermoConstants
        tc = ThermoConstants()
        self._sa = _OGSA(tc.Ll, tc.cl, tc.T_energy, tc.Ttr, tc.ptr, float(sa_abstol), int(sa_maxiter))
        zc_halo = np.zeros(g.Szc)
        zc_halo[g.Hz:g.Hz + g.Nz] = g.zc
        self._zc_halo = zc_halo
        # fields
        self.ru, self.rv, self.rtheta, self.rq = (g.center_field() for _ in range(4))
        self.rw = g.zface_field()
        self.u, self.v, self.theta, self.q, self.T, self.phi = (g.center_field() for _ in range(6))
        self.w = g.zface_field()
        # user tracers (tracers = (:a, :b)): prognostic rho c ("rc0", "rc1", ...), specific c ("c0", ...)
        self.n_tracers = int(tracers)
        if self.n_tracers:
            self.PROGNOSTIC = tuple(self.PROGNOSTIC) + tuple(f"rc{t}" for t in range(self.n_tracers))
            for t in range(self.n_tracers):
                setattr(self, f"rc{t}", g.center_field())
                setattr(self, f"c{t}", g.center_field())
        self.U0 = {n: np.zeros_like(getattr(self, n)) for n in self.PROGNOSTIC}
        self.G = {n: np.zeros_like(getattr(self, n)) for n in self.PROGNOSTIC}
        # Poisson solver setup (dynamics_pressure_solver, anelastic_pressure_solver.jl:11-24)
        self.lower = np.zeros(max(g.Nz - 1, 1))
        self.diag0 = np.zeros(g.Nz)
        self.mass = np.zeros(g.Nz)
        self.lib.og_poisson_coefficients(C.byref(self.cg), _p(self.lower), _p(self.diag0), _p(self.mass))
        self.clock_time, self.iteration = 0.0, 0
        if g.topo[1] != SLAB:
            lx = poisson_eigenvalues(g.Nx, g.dx, g.topo[0])
            ly = poisson_eigenvalues(g.Ny, g.dy, g.topo[1])
            self.lam = np.ascontiguousarray(ly[:, None] + lx[None, :])
        if initialize:
            # initialize_model_thermodynamics!: theta = theta0 (anelastic_time_stepping.jl:15-19)
            self.set(theta=self.ref.theta0)

    def _mk_cgrid(self):
        g, c, r = self.grid, self.constants, self.ref
        self.cg = _OGGrid(g.Nx, g.Ny, g.Nz, g.Hx, g.Hy, g.Hz, g.topo[0], g.topo[1], g.topo[2],
                          g.dx, g.dy, _p(g.dzc), _p(g.dzf), _p(r.density), _p(r.pressure),
                          _p(r.temperature), c.g, c.Rd, c.Rv, c.cpd, c.cpv, r.pst)

    # -- halo filling -------------------------------------------------------
    def _halo_xy(self, f, yface=False, xface=False):
        # x / y of every field: periodic wrap; a Bounded direction adds the no-flux cell / row (centres) or the impenetrable wall faces
        cg = C.byref(self.cg)
        self.lib.og_fill_halo_periodic_xy(cg, _p(f), C.c_int(f.shape[0]))
        if self.grid.topo[0] == BOUNDED:
            (self.lib.og_fill_halo_x_wall if xface else self.lib.og_fill_halo_x_noflux)(cg, _p(f), C.c_int(f.shape[0]))
        if self.grid.topo[1] == BOUNDED:
            (self.lib.og_fill_halo_y_wall if yface else self.lib.og_fill_halo_y_noflux)(cg, _p(f), C.c_int(f.shape[0]))

    def _halo_center(self, f, yface=False, xface=False):
        self._halo_xy(f, yface, xface)
        if self.grid.Hz > 0:
            self.lib.og_fill_halo_z_noflux(C.byref(self.cg), _p(f))

    def _halo_w(self, f, wall=True):
        self._halo_xy(f)
        if wall:
            self.lib.og_fill_halo_z_wall(C.byref(self.cg), _p(f))

    def _halo_velocity(self, f, yface=False, xface=False):   # `nothing` BC in z: periodic wrap only (anelastic_dynamics.jl:174-182)
        self._halo_xy(f, yface, xface)

    def fill_momentum_halos(self):
        self._halo_center(self.ru, xface=True)
        self._halo_center(self.rv, yface=True)
        self._halo_w(self.rw)

    # -- set! ----------------------------------------------------------------
    def _eval(self, value, loc):
        g = self.grid
        if callable(value):
            x, y, z = g.nodes(loc)
            out = value(x, y, z)
            shape = (g.Nz + (1 if loc[2] == "f" else 0), g.Ny, g.Nx)
            return np.broadcast_to(np.asarray(out, dtype=np.float64), shape)
        return value

    def set(self, enforce_mass_conservation=True, **kw):
        g = self.grid
        Hz, Nz = g.Hz, g.Nz
        rho_c = self.ref.density[Hz:Hz + Nz][:, None, None]
        rho_f = (0.5 * (self.ref.density[Hz - 1:Hz + Nz] + self.ref.density[Hz:Hz + Nz + 1]))[:, None, None] \
            if Hz > 0 else None
        for name, value in sorted(kw.items(), key=lambda kv: 0 if kv[0] in ("qt", "qv", "rq") else 1):   # moisture first
            if name in ("qt", "qv"):
                g.interior(self.q)[...] = self._eval(value, "ccc")
                g.interior(self.rq)[...] = rho_c * g.interior(self.q)
            elif name == "rq":
                g.interior(self.rq)[...] = self._eval(value, "ccc")
            elif name in ("qcl", "qr") and self.microphysics == "Kessler":
                # settable specific microphysical names (set_atmosphere_model.jl:247-253): only rho*q is set; the diagnostic
                # q^cl / q^r field is refreshed by the update_state! that follows (after T was diagnosed from the old one)
                g.interior(getattr(self, "r" + name))[...] = rho_c * self._eval(value, "ccc")
            elif name == "u":
                g.interior(self.u)[...] = self._eval(value, "fcc")
                g.interior(self.ru)[...] = rho_c * g.interior(self.u)
            elif name == "v":
                g.interior(self.v)[...] = self._eval(value, "cfc")
                g.interior(self.rv)[...] = rho_c * g.interior(self.v)
            elif name == "w":
                g.interior(self.w, True)[...] = self._eval(value, "ccf")
                g.interior(self.rw, True)[...] = rho_f * g.interior(self.w, True)
            elif name == "ru":
                g.interior(self.ru)[...] = self._eval(value, "fcc")
            elif name == "rv":
                g.interior(self.rv)[...] = self._eval(value, "cfc")
            elif name == "rw":
                g.interior(self.rw, True)[...] = self._eval(value, "ccf")
            elif name == "theta" and self.formulation == "StaticEnergy":
                # _energy_density_from_potential_temperature! (static_energy_tendency.jl:93-140): T = Pi theta,
                # e = cpm T + g z - 0 - 0, rho_e = rho e
                c, r = self.constants, self.ref
                th = np.asarray(self._eval(value, "ccc"), dtype=np.float64)
                q = g.interior(self.q)
                qd = 1.0 - (q + 0.0 + 0.0)
                Rm = qd * c.Rd + q * c.Rv
                cpm = qd * c.cpd + q * c.cpv + 0.0 + 0.0
                pr = r.pressure[Hz:Hz + Nz][:, None, None]
                T = (pr / r.pst) ** (Rm / cpm) * th + 0.0
                e = cpm * T + c.g * g.zc[:, None, None] - 0.0 - 0.0
                g.interior(self.theta)[...] = e
                g.interior(self.rtheta)[...] = rho_c * e
            elif name in ("theta", "e"):
                g.interior(self.theta)[...] = self._eval(value, "ccc")
                g.interior(self.rtheta)[...] = rho_c * g.interior(self.theta)
            elif name == "re":
                g.interior(self.rtheta)[...] = self._eval(value, "ccc")
            elif name == "rtheta":
                g.interior(self.rtheta)[...] = self._eval(value, "ccc")
            elif name.startswith("rc") and name[2:].isdigit() and int(name[2:]) < self.n_tracers:
                g.interior(getattr(self, name))[...] = self._eval(value, "ccc")
            else:
                raise ValueError(f"Cannot set {name} in OracleModel")
        self.update_state(compute_tendencies=False)
        if enforce_mass_conservation:        # set_atmosphere_model.jl:121-128
            self.compute_pressure_correction(1.0)
            self.make_pressure_correction(1.0)
            self.update_state(compute_tendencies=False)

    # -- update_state! --------------------------------------------------------
    def update_state(self, compute_tendencies=True):
        cg = C.byref(self.cg)
        self.fill_momentum_halos()
        self._halo_center(self.rtheta)
        self._halo_center(self.rq)
        self.lib.og_compute_velocities(cg, _p(self.u), _p(self.v), _p(self.w), _p(self.ru), _p(self.rv), _p(self.rw))
        for f in (self.u, self.v, self.w):
            self._halo_velocity(f, yface=f is self.v, xface=f is self.u)
        if self.microphysics == "Kessler":
            # microphysical_state + grid_moisture_fractions + update_microphysical_auxiliaries! (dcmip2016_kessler.jl:222-227,
            # 298-303,860-865): q = (q^v, q^cl + q^r), T = Pi(q) theta + L q^l / c_pm
            self._halo_center(self.rqcl)
            self._halo_center(self.rqr)
            g, c, t = self.grid, self.constants, self.tetens
            I = g.interior
            rho = self.ref.density[g.Hz:g.Hz + g.Nz][:, None, None]
            pr = self.ref.pressure[g.Hz:g.Hz + g.Nz][:, None, None]
            th, qv = I(self.rtheta) / rho, I(self.rq) / rho
            # NB (reference behaviour, update_atmosphere_model_state.jl:276-291): grid_moisture_fractions reads the
            # *diagnostic* fields mu.q^cl, mu.q^r before update_microphysical_auxiliaries! refreshes them in the same kernel,
            # so the temperature carries the condensate of the previous update_state!
            ql_lag = I(self.qcl) + I(self.qr)
            qd = 1.0 - (qv + ql_lag + 0.0)
            Rm = qd * c.Rd + qv * c.Rv
            cpm = qd * c.cpd + qv * c.cpv + ql_lag * t.cl + 0.0
            I(self.T)[...] = (pr / self.ref.pst) ** (Rm / cpm) * th + (t.Ll * ql_lag + 0.0) / cpm
            qcl, qr = I(self.rqcl) / rho, I(self.rqr) / rho
            I(self.theta)[...], I(self.q)[...], I(self.qcl)[...], I(self.qr)[...] = th, qv, qcl, qr
            I(self.qv)[...], I(self.ql)[...] = qv, qcl + qr
            for f in (self.qcl, self.qr, self.qv, self.ql):
                self._halo_center(f)
        elif self.microphysics == "SaturationAdjustment":
            self.lib.og_compute_thermo_sa(cg, C.byref(self._sa), _p(self.theta), _p(self.q), _p(self.qv), _p(self.ql),
                                          _p(self.T), _p(self.rtheta), _p(self.rq))
            self._halo_center(self.qv)
            self._halo_center(self.ql)
        elif self.formulation == "StaticEnergy":
            self.lib.og_compute_thermo_energy(cg, _p(self.theta), _p(self.q), _p(self.T), _p(self.rtheta), _p(self.rq),
                                              _p(self._zc_halo))
        else:
            self.lib.og_compute_thermo(cg, _p(self.theta), _p(self.q), _p(self.T), _p(self.rtheta), _p(self.rq))
        for f in (self.T, self.q, self.theta):
            self._halo_center(f)
        if self.n_tracers:        # tracer_density_to_specific! (update_atmosphere_model_state.jl:43,97-103) + halo fill
            g = self.grid
            rho = self.ref.density[g.Hz:g.Hz + g.Nz][:, None, None]
            for t in range(self.n_tracers):
                c = getattr(self, f"c{t}")
                g.interior(c)[...] = g.interior(getattr(self, f"rc{t}")) / rho
                self._halo_center(c)
        if self.closure is not None:       # compute_closure_fields! closes compute_auxiliary_variables! (:218)
            from .closure import compute_closure_fields
            compute_closure_fields(self)
        if self.forcings is not None:
            from .forcings import compute_forcings
            compute_forcings(self)
        if compute_tendencies:
            self.compute_tendencies()

    def compute_tendencies(self):
        # the WENO order is a process-wide switch of the C library: select this model's order for the evaluation and put the
        # default back afterwards, so that direct callers of og_*_tendency (tests) always see order 5
        self.lib.og_set_weno_order(C.c_int(self.weno_order))
        self.lib.og_set_weno_ft2(C.c_int(self.weno_ft2))
        try:
            self._compute_tendencies()
        finally:
            self.lib.og_set_weno_order(C.c_int(5))
            self.lib.og_set_weno_ft2(C.c_int(0))

    def _compute_tendencies(self):
        cg = C.byref(self.cg)
        L, G = self.lib, self.G
        L.og_u_tendency(cg, _p(G["ru"]), _p(self.ru), _p(self.rv), _p(self.rw), _p(self.u))
        L.og_v_tendency(cg, _p(G["rv"]), _p(self.ru), _p(self.rv), _p(self.rw), _p(self.v))
        if self.microphysics == "Kessler":
            L.og_set_weno_order(C.c_int(self.scalar_order))
            L.og_scalar_tendency(cg, _p(G["rqcl"]), _p(self.u), _p(self.v), _p(self.w), _p(self.qcl))
            L.og_scalar_tendency(cg, _p(G["rqr"]), _p(self.u), _p(self.v), _p(self.w), _p(self.qr))
            L.og_set_weno_order(C.c_int(self.weno_order))
        if self.microphysics in ("SaturationAdjustment", "Kessler"):
            L.og_w_tendency_moist(cg, _p(G["rw"]), _p(self.ru), _p(self.rv), _p(self.rw), _p(self.w), _p(self.T),
                                  _p(self.qv), _p(self.ql))
        else:
            L.og_w_tendency(cg, _p(G["rw"]), _p(self.ru), _p(self.rv), _p(self.rw), _p(self.w), _p(self.T), _p(self.q))
        L.og_set_weno_order(C.c_int(self.scalar_order))      # every scalar below takes the scalars' scheme
        L.og_scalar_tendency(cg, _p(G["rtheta"]), _p(self.u), _p(self.v), _p(self.w), _p(self.theta))
        L.og_scalar_tendency(cg, _p(G["rq"]), _p(self.u), _p(self.v), _p(self.w), _p(self.q))
        if self.formulation == "StaticEnergy":
            L.og_energy_buoyancy_flux(cg, _p(G["rtheta"]), _p(self.w), _p(self.T), _p(self.q))
        for t in range(self.n_tracers):     # scalar_tendency per tracer (update_atmosphere_model_state.jl:352-372)
            L.og_scalar_tendency(cg, _p(G[f"rc{t}"]), _p(self.u), _p(self.v), _p(self.w), _p(getattr(self, f"c{t}")))
        # advection = (; rho_q = WENO(order = 5, bounds = (lo, hi))) (examples/rico.jl:184-190): bounds-preserving divergence
        for key, (lo, hi) in getattr(self, "bounded", {}).items():
            specific = {"rq": self.q, "rqcl": getattr(self, "qcl", None), "rqr": getattr(self, "qr", None)}.get(key)
            if key.startswith("rc"):
                specific = getattr(self, key[1:])
            L.og_scalar_tendency_bounded(cg, _p(G[key]), _p(self.u), _p(self.v), _p(self.w), _p(specific),
                                         C.c_double(lo), C.c_double(hi))
        if self.closure is not None:
            from .closure import add_closure_tendencies
            add_closure_tendencies(self)
        if self.forcings is not None:
            from .forcings import add_forcing_tendencies
            add_forcing_tendencies(self)
        if getattr(self, "relaxation", None):
            from .forcings import add_relaxation_tendencies
            add_relaxation_tendencies(self)
        if getattr(self, "field_forcing", None) is not None:
            from .forcings import add_field_forcing
            add_field_forcing(self)

    def liquid_ice_potential_temperature(self):
        """Diagnostics.LiquidIcePotentialTemperature (dry: theta = T / Pi) on the interior."""
        g, c, r = self.grid, self.constants, self.ref
        if self.formulation != "StaticEnergy":
            return g.interior(self.theta).copy()
        q = g.interior(self.q)
        qd = 1.0 - q
        kappa = (qd * c.Rd + q * c.Rv) / (qd * c.cpd + q * c.cpv)
        Pi = (r.pressure[g.Hz:g.Hz + g.Nz][:, None, None] / r.pst) ** kappa
        return g.interior(self.T) / Pi

    # -- pressure ------------------------------------------------------------
    def solve_poisson(self, rhs):
        """solve!(phi, FourierTridiagonalPoissonSolver) (Oceananigans, recalled):
        FFT x,y -> complex Thomas in z -> inverse FFT -> subtract mean -> real part."""
        g = self.grid
        workers = getattr(self, "fft_workers", 1)
        if BOUNDED in (g.topo[0], g.topo[1]):       # cosine transform along a Bounded direction (DCT-II forward, its inverse back), FFT along a periodic one
            import scipy.fft as sfft

            def along(a, axis, topo, inverse):
                if topo == BOUNDED:
                    t = sfft.idct if inverse else sfft.dct
                    return t(a.real, type=2, axis=axis) + 1j * t(a.imag, type=2, axis=axis)
                return (sfft.ifft if inverse else sfft.fft)(a, axis=axis)      # Flat: a length-1 transform
            fft2 = lambda a: along(along(a, 1, g.topo[1], False), 2, g.topo[0], False)
            ifft2 = lambda a: along(along(a, 2, g.topo[0], True), 1, g.topo[1], True)
        elif workers > 1:       # bench.py's cpu_baseline leg: the same pocketfft transforms on several host threads
            import scipy.fft as sfft
            fft2 = lambda a: sfft.fft2(a, axes=(1, 2), workers=workers)
            ifft2 = lambda a: sfft.ifft2(a, axes=(1, 2), workers=workers)
        else:
            fft2 = lambda a: np.fft.fft2(a, axes=(1, 2))
            ifft2 = lambda a: np.fft.ifft2(a, axes=(1, 2))
        rhat = np.ascontiguousarray(fft2(rhs.astype(np.complex128)))
        phat = np.zeros_like(rhat)
        scratch = np.zeros(rhs.shape)
        self.lib.og_tridiagonal_solve(C.c_int(g.Nx), C.c_int(g.Ny), C.c_int(g.Nz), _p(self.lower),
                                      _p(self.diag0), _p(self.mass), _p(self.lam),
                                      rhat.view(np.float64).ctypes.data_as(_dp),
                                      phat.view(np.float64).ctypes.data_as(_dp), _p(scratch))
        phi = ifft2(phat)
        phi = phi - phi.mean()
        return np.ascontiguousarray(phi.real)

    def compute_pressure_correction(self, dt):
        g = self.grid
        self.fill_momentum_halos()
        rhs = np.zeros((g.Nz, g.Ny, g.Nx))
        self.lib.og_poisson_source(C.byref(self.cg), _p(rhs), _p(self.ru), _p(self.rv), _p(self.rw), C.c_double(dt))
        g.interior(self.phi)[...] = self.solve_poisson(rhs)
        self._halo_center(self.phi)

    def make_pressure_correction(self, dt):
        self.lib.og_pressure_correct(C.byref(self.cg), _p(self.ru), _p(self.rv), _p(self.rw), _p(self.phi), C.c_double(dt))

    def divergence(self):
        g = self.grid
        d = np.zeros((g.Nz, g.Ny, g.Nx))
        self.lib.og_divergence(C.byref(self.cg), _p(d), _p(self.ru), _p(self.rv), _p(self.rw))
        return d

    # -- time stepping -------------------------------------------------------
    def rk3_substep(self, dt, alpha):
        g = self.grid
        for n in self.PROGNOSTIC:
            k0, k1 = (1, g.Nz) if n == "rw" else (0, g.Nz)
            self.lib.og_rk3_substep(C.byref(self.cg), _p(getattr(self, n)), _p(self.U0[n]), _p(self.G[n]),
                                    C.c_double(dt), C.c_double(alpha), C.c_int(k0), C.c_int(k1))

    def time_step(self, dt):
        if self.iteration == 0:
            self.update_state(compute_tendencies=True)
        for n in self.PROGNOSTIC:                      # store_initial_state!
            self.U0[n][...] = getattr(self, n)
        for alpha in (1.0, 1.0 / 4.0, 2.0 / 3.0):
            if self.forcings is not None:              # compute_flux_bc_tendencies! (ssp_runge_kutta_3.jl:229,243,257)
                from .forcings import add_flux_bc_tendencies
                add_flux_bc_tendencies(self)
            self.rk3_substep(dt, alpha)
            self.compute_pressure_correction(alpha * dt)
            self.make_pressure_correction(alpha * dt)
            self.update_state(compute_tendencies=True)
        if self.microphysics == "Kessler":
            self.microphysics_model_update(dt)
        self.clock_time += dt
        self.iteration += 1

    def microphysics_model_update(self, dt):
        """microphysics_model_update!(::DCMIP2016KesslerMicrophysics, model) (dcmip2016_kessler.jl:449-486): the column
        kernel on every column, then update_state!."""
        from .kessler import kessler_column_update
        g = self.grid
        I = g.interior
        rho = np.ascontiguousarray(self.ref.density[g.Hz:g.Hz + g.Nz])
        p = np.ascontiguousarray(self.ref.pressure[g.Hz:g.Hz + g.Nz])
        for j in range(g.Ny):
            for i in range(g.Nx):
                cols = [np.ascontiguousarray(I(f)[:, j, i]) for f in (self.theta, self.rtheta, self.rq, self.rqcl, self.rqr)]
                qv, qcl, qr, W, P, _ = kessler_column_update(dt, rho, p, self.ref.pst, g.zc, *cols, self.kessler, self.tetens)
                for f, col in zip((self.theta, self.rtheta, self.rq, self.rqcl, self.rqr), cols):
                    I(f)[:, j, i] = col
                I(self.q)[:, j, i], I(self.qv)[:, j, i], I(self.qcl)[:, j, i], I(self.qr)[:, j, i], I(self.W)[:, j, i] = qv, qv, qcl, qr, W
                self.precipitation_rate[j, i] = P
        self.update_state(compute_tendencies=True)
