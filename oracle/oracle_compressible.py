"""
oracle_compressible.py — CPU (numpy + C) restatement of Breeze.jl's compressible
split-explicit time step.

TEST INFRASTRUCTURE ONLY.  Imported by tests/ (and nothing in breeze.jl_amd/).

PARITY STATUS: Breeze-side arithmetic follows the cited lines; pinned by the reference's own
known-answer tests restated in tests/test_oracle_compressible.py.  WENO arithmetic of the slow
tendencies: "parity unpinned" (see breeze_oracle.c header).

Reference call stack restated here (file:line relative to /root/reference):
  time_step!                    src/TimeSteppers/acoustic_runge_kutta_3.jl:264-319
  acoustic_rk3_substep!         src/TimeSteppers/acoustic_runge_kutta_3.jl:172-192
  compute_slow_*_tendencies!    src/TimeSteppers/acoustic_substep_helpers.jl:55-149
  scalar_substep!               src/TimeSteppers/acoustic_substep_helpers.jl:197-248
  acoustic_rk3_substep_loop!    src/CompressibleEquations/acoustic_substepping.jl:1404-1590
  refresh_linearization_basic_state!                         ...:318-399
  compute_acoustic_substeps / stage_substep_count_and_size   ...:451-508
  update_state!                 src/AtmosphereModels/update_atmosphere_model_state.jl:41-68
  _compute_temperature_and_pressure!  src/CompressibleEquations/compressible_time_stepping.jl:191-242
  ExnerReferenceState           src/Thermodynamics/reference_states.jl:588-672,717-815
"""
import ctypes as C
import math

import numpy as np

from .oracle import BOUNDED, FLAT, Constants, Grid, _OGGrid, _dp, _p, lib  # noqa: F401


# ---------------------------------------------------------------------------
# ExnerReferenceState: dry, 1-D column, isentropic mode (reference_states.jl:588-672)
# ---------------------------------------------------------------------------
class ExnerReferenceState:
    """ExnerReferenceState(grid, constants; surface_pressure, potential_temperature, standard_pressure, vapor_mass_fraction)
    — isentropic mode, dry or moist (src/Thermodynamics/reference_states.jl:572-577 moist_reference_constants, :588-598 Newton on
    the discrete balance, :609-672 _compute_exner_column! / integrate_exner_column!, :718-827 constructor)."""

    def __init__(self, grid, constants, surface_pressure=101325.0, potential_temperature=288.0,
                 standard_pressure=1e5, vapor_mass_fraction=None):
        g, c = grid, constants
        Nz, Hz = g.Nz, g.Hz
        self.p0, self.pst = float(surface_pressure), float(standard_pressure)
        th_fun = potential_temperature if callable(potential_temperature) else (lambda z: float(potential_temperature) + 0.0 * z)
        self.theta0 = float(th_fun(np.float64(0.0)))
        theta = np.zeros(g.Szc)
        theta[Hz:Hz + Nz] = th_fun(g.zc)
        theta[Hz - 1] = theta[Hz]
        theta[Hz + Nz] = theta[Hz + Nz - 1]
        if vapor_mass_fraction is None:
            qv = np.zeros(Nz)
        elif callable(vapor_mass_fraction):
            qv = np.array([float(vapor_mass_fraction(z)) for z in g.zc])
        else:
            qv = np.full(Nz, float(vapor_mass_fraction))
        self.vapor_mass_fraction = qv

        def moist(q):          # moist_reference_constants
            qd = 1 - q
            Rm = qd * c.Rd + q * c.Rv
            cpm = qd * c.cpd + q * c.cpv
            return Rm, cpm, Rm / cpm

        grav, pst = c.g, self.pst
        pi = np.zeros(g.Szc)
        p = np.zeros(g.Szc)
        rho = np.zeros(g.Szc)
        Rm1, cpm1, kap1 = moist(qv[0])
        pi_surf = (self.p0 / pst) ** kap1
        th1 = theta[Hz]
        Pi1 = pi_surf - grav * g.dzc[Hz] / (2 * cpm1 * th1)
        p1 = pst * Pi1 ** (1 / kap1)
        pi[Hz], p[Hz], rho[Hz] = Pi1, p1, p1 / (Rm1 * th1 * Pi1)
        pm, rm = p[Hz], rho[Hz]
        for k in range(1, Nz):
            dzf = g.dzf[k + Hz]
            thk, thm = theta[Hz + k], theta[Hz + k - 1]
            Rm, cpm, kap = moist(qv[k])
            thf = (thk + thm) / 2
            Pi_init = pi[Hz + k - 1] - grav * dzf / (cpm * thf)
            pk = pst * Pi_init ** (1 / kap)
            A = grav * pst ** kap / (2 * Rm * thk)
            Cc = pm / dzf - grav * rm / 2
            for _ in range(5):                  # FixedIterations(5)
                rp = pk ** (-kap)
                f = pk / dzf + A * pk * rp - Cc
                fp = 1 / dzf + A * (1 - kap) * rp
                pk -= f / fp
            Pik = (pk / pst) ** kap
            rk = pk / (Rm * thk * Pik)
            pi[Hz + k], p[Hz + k], rho[Hz + k] = Pik, pk, rk
            pm, rm = pk, rk
        self.rho0 = self.p0 / (Rm1 * self.theta0 * pi_surf)
        # halos: bottom Value BC for p, rho; zero-gradient elsewhere (first halo cell)
        p[Hz - 1] = 2 * self.p0 - p[Hz]
        rho[Hz - 1] = 2 * self.rho0 - rho[Hz]
        pi[Hz - 1] = pi[Hz]
        for a in (p, rho, pi):
            a[Hz + Nz] = a[Hz + Nz - 1]
        self.pressure, self.density, self.exner_function, self.potential_temperature = p, rho, pi, theta


# ---------------------------------------------------------------------------
# SplitExplicitTimeDiscretization defaults (time_discretizations.jl:550-562)
# ---------------------------------------------------------------------------
class SplitExplicit:
    def __init__(self, substeps=None, acoustic_cfl=0.5, forward_weight=0.65,
                 damping_coefficient=0.1, damp_vertical=False,
                 apply_first_substep_pressure_gradient=False,
                 thermodynamic_tendency_factor=1.0, vertical_momentum_tendency_factor=1.0, direct_damping=False, sponge=None,
                 substep_distribution="proportional", damping_length_scale=None):
        self.damping_length_scale = None if damping_length_scale is None else float(damping_length_scale)   # ThermalDivergenceDamping(length_scale)
        self.direct_damping = bool(direct_damping)      # DirectDivergenceDamping(coefficient) instead of ThermalDivergenceDamping
        self.sponge = sponge                            # None or (damping_rate, depth, ramp) with ramp in {"linear", "cubic", "sin2"}
        assert substep_distribution in ("proportional", "constant", "monolithic_first_stage")
        self.substep_distribution = substep_distribution
        self.substeps = substeps
        self.acoustic_cfl = float(acoustic_cfl)
        self.forward_weight = float(forward_weight)
        self.damping_coefficient = None if damping_coefficient is None else float(damping_coefficient)  # None = NoDivergenceDamping
        self.damp_vertical = bool(damp_vertical)
        self.apply_first = bool(apply_first_substep_pressure_gradient)
        self.f_theta = float(thermodynamic_tendency_factor)
        self.f_w = float(vertical_momentum_tendency_factor)


def upper_sponge_profile(zf, Lz, damping_rate=0.2, depth=5e3, ramp="cubic"):
    """damping_rate * ramp(z, Lz, depth) on the faces zf (time_discretizations.jl:398-433: LinearRamp, CubicRamp, Sin2Ramp; the
    reference passes grid.Lz as the sponge top; UpperSponge defaults :497)."""
    s = np.clip((np.asarray(zf, dtype=np.float64) - (Lz - depth)) / depth, 0.0, 1.0)
    shape = {"linear": s, "cubic": s * s * (3 - 2 * s), "sin2": np.sin(np.pi / 2 * s) ** 2}[ramp]
    return np.ascontiguousarray(damping_rate * shape)


def compute_acoustic_substeps(grid, dt, constants, acoustic_cfl):
    """acoustic_substepping.jl:451-466"""
    c = constants
    gam = c.cpd / (c.cpd - c.Rd)
    cs = math.sqrt(gam * c.Rd * 300.0)
    dx = math.inf if grid.topo[0] == FLAT else grid.dx
    dy = math.inf if grid.topo[1] == FLAT else grid.dy
    return max(1, math.ceil(abs(float(dt)) * cs / (acoustic_cfl * min(dx, dy))))


def stage_substep_count_and_size(substeps, beta, dt, grid, constants, acoustic_cfl, distribution="proportional"):
    """ProportionalSubsteps (acoustic_substepping.jl:491-495), ConstantSubstepSize (:497-501), MonolithicFirstStage (:503-508)"""
    if distribution == "monolithic_first_stage" and beta < (1 / 3 + 1 / 2) / 2:
        return 1, dt / 3
    if distribution in ("constant", "monolithic_first_stage"):
        n_raw = substeps if substeps is not None else compute_acoustic_substeps(grid, dt, constants, acoustic_cfl)
        N = max(6, 6 * -(-n_raw // 6))          # _uniform_substep_count (:484-487)
        return max(1, round(beta * N)), dt / N
    dt_stage = beta * dt
    if substeps is None:
        n = compute_acoustic_substeps(grid, dt_stage, constants, acoustic_cfl)
    else:
        n = max(1, math.ceil(beta * substeps))
    return n, dt_stage / n


def apply_horizontal_pressure_gradient_substep(substep, n_tau, apply_first=False):
    """acoustic_substepping.jl:891-895 (substep is 1-based)"""
    return bool(apply_first) or (substep != 1) or (n_tau == 1)


class CompressibleOracleModel:
    """AtmosphereModel(grid; advection=WENO(order=5), dynamics=CompressibleDynamics(
    SplitExplicitTimeDiscretization(...); reference_potential_temperature=...)) on the CPU."""

    PROGNOSTIC = ("rho_d", "ru", "rv", "rw", "rtheta", "rq")
    BETAS = (1.0 / 3.0, 1.0 / 2.0, 1.0)

    def __init__(self, grid, constants=None, time_discretization=None, surface_pressure=101325.0,
                 standard_pressure=1e5, reference_potential_temperature=288.0, reference_state=True,
                 newton_abstol=1e-4, newton_maxiter=8, microphysics=None, reference_vapor_mass_fraction=None,
                 advection="WENO5", coriolis_f=0.0, relaxation=None):
        # coriolis_f: FPlane(f); relaxation: {"ru" | "rv" | "rw" | "rtheta": (rate column, target column)} — the sponge layers of
        # examples/tropical_cyclone_with_rainband.jl:434-514 (oracle/forcings.py: add_relaxation_tendencies); both are slow terms
        self.coriolis_f, self.relaxation = float(coriolis_f), relaxation
        # microphysics "Kessler": DCMIP2016KesslerMicrophysics — rho q^cl, rho q^r prognostic (dcmip2016_kessler.jl:216)
        # "SaturationAdjustment": SaturationAdjustment(equilibrium = WarmPhaseEquilibrium()) on the density-based state
        # (saturation_adjustment.jl:236-301); self.q / self.rq hold the total (equilibrium) moisture, self.qv / self.ql the partition
        assert microphysics in (None, "Kessler", "SaturationAdjustment")
        self.microphysics = microphysics
        self.grid = g = grid
        self.constants = c = constants or Constants()
        self.td = time_discretization or SplitExplicit()
        self.pst = float(standard_pressure)
        self.p0 = float(surface_pressure)
        self.ref = (ExnerReferenceState(g, c, surface_pressure, reference_potential_temperature, standard_pressure,
                                        vapor_mass_fraction=reference_vapor_mass_fraction)
                    if reference_state else None)
        self.newton = (float(newton_abstol), int(newton_maxiter))
        # advection "WENO7" / "WENO9" (examples/splitting_supercell.jl:279): the order is a process-wide switch of the C library, selected
        # around every tendency evaluation and put back to 5 afterwards (oracle.py does the same)
        self.weno_order = {"WENO5": 5, "WENO7": 7, "WENO9": 9}.get(advection, 5)
        if self.weno_order != 5:
            need = (self.weno_order + 1) // 2
            assert min(h for h, t in zip((g.Hx, g.Hy, g.Hz), g.topo) if t != FLAT) >= need, "halo too narrow for this WENO order"
        self.lib = lib("WENO5" if self.weno_order != 5 else advection)
        zeros = np.zeros(g.Szc)
        self._zc = zeros
        self.cg = _OGGrid(g.Nx, g.Ny, g.Nz, g.Hx, g.Hy, g.Hz, g.topo[0], g.topo[1], g.topo[2],
                          g.dx, g.dy, _p(g.dzc), _p(g.dzf), _p(zeros), _p(zeros), _p(zeros),
                          c.g, c.Rd, c.Rv, c.cpd, c.cpv, self.pst)
        cf, zf = g.center_field, g.zface_field
        self.rho_d, self.rho, self.rtheta, self.rq, self.ru, self.rv = (cf() for _ in range(6))
        self.rw = zf()
        self.u, self.v, self.theta, self.q, self.T, self.p = (cf() for _ in range(6))
        self.w = zf()
        if microphysics == "Kessler":
            from .kessler import KesslerParameters, TetensConstants
            self.PROGNOSTIC = CompressibleOracleModel.PROGNOSTIC + ("rqcl", "rqr")
            self.rqcl, self.rqr, self.qcl, self.qr, self.W = (cf() for _ in range(5))
            self.precipitation_rate = np.zeros((g.Ny, g.Nx))
            self.kessler = KesslerParameters()
            self.tetens = TetensConstants(molar_gas_constant=c.R, dry_air_molar_mass=c.Md, vapor_molar_mass=c.Mv,
                                          dry_air_heat_capacity=c.cpd, vapor_heat_capacity=c.cpv)
        if microphysics == "SaturationAdjustment":
            from .thermo import ThermoConstants
            self.qv, self.ql = cf(), cf()
            self.thermo = ThermoConstants()
            self.sa_solver = (1e-4, 20)            # SecantSolver(abstol, maxiter) (saturation_adjustment.jl:55-59)
        self.U0 = {n: np.zeros_like(getattr(self, n)) for n in self.PROGNOSTIC}
        self.G = {n: np.zeros_like(getattr(self, n)) for n in self.PROGNOSTIC}
        # AcousticSubstepper storage (acoustic_substepping.jl:207-231)
        self.Pi, self.thL, self.gR = cf(), cf(), cf()
        self.rp, self.rthp, self.rup, self.rvp = cf(), cf(), cf(), cf()
        self.rwp = zf()
        self.rs, self.rths, self.rth_old = cf(), cf(), cf()
        self.au, self.av, self.aw = cf(), cf(), zf()
        self.Gs, self.rhs = zf(), zf()
        self.iteration, self.clock_time = 0, 0.0
        self.last_substeps = []
        # sides of a Bounded x / y with an active open (normal-flow) condition on the wall-normal momentum; SplitExplicitTimeDiscretization(
        # open_boundary_relaxation = 0.5) (time_discretizations.jl:562)
        self.lateral_open = dict(west=False, east=False, south=False, north=False)
        self.open_boundary_relaxation = 0.5
        # seed_pressure! (compressible_dynamics.jl:254-258)
        if self.ref is not None:
            g.interior(self.p)[...] = self.ref.pressure[g.Hz:g.Hz + g.Nz][:, None, None]
        else:
            g.interior(self.p)[...] = self.p0
        self._halo_center(self.p)

    # -- halos ---------------------------------------------------------------
    def _halo_center(self, f):
        self.lib.og_fill_halo_periodic_xy(C.byref(self.cg), _p(f), C.c_int(f.shape[0]))
        self.lib.og_fill_halo_z_noflux(C.byref(self.cg), _p(f))

    # -- lateral walls / open boundaries (round 6): the substepper's own fields carry the default boundary conditions of their location
    # (acoustic_substepping.jl:207-231; the reference's test/acoustic_substepping_open_boundaries.jl:70-73 pins `west === nothing` on the
    # momentum perturbation): zero-gradient halo along a Bounded direction for a field that is a centre there, wall faces untouched
    def _halo_sub(self, f, xface=False, yface=False):
        self._halo_center(f)      # periodic directions and z (y-slab tests replace this method by their exchange)
        n = C.c_int(f.shape[0])
        cg = C.byref(self.cg)
        if self.grid.topo[0] == BOUNDED and not xface:
            self.lib.og_fill_halo_x_noflux(cg, _p(f), n)
        if self.grid.topo[1] == BOUNDED and not yface:
            self.lib.og_fill_halo_y_noflux(cg, _p(f), n)

    def _walls(self):
        return BOUNDED in (self.grid.topo[0], self.grid.topo[1])

    def enforce_wall_impenetrability(self):
        """enforce_wall_impenetrability! (acoustic_substepping.jl:1378-1395): the wall-normal momentum perturbation on the two wall faces of
        a Bounded direction whose momentum boundary condition is the default (impenetrable) one; self.lateral_open names the sides that
        carry an active open boundary condition instead"""
        g, cg, op = self.grid, C.byref(self.cg), self.lateral_open
        if g.topo[0] == BOUNDED:
            if not op["west"]:
                self.lib.og_zero_wall_face(cg, _p(self.rup), C.c_int(0), C.c_int(0))
            if not op["east"]:
                self.lib.og_zero_wall_face(cg, _p(self.rup), C.c_int(0), C.c_int(g.Nx))
        if g.topo[1] == BOUNDED:
            if not op["south"]:
                self.lib.og_zero_wall_face(cg, _p(self.rvp), C.c_int(1), C.c_int(0))
            if not op["north"]:
                self.lib.og_zero_wall_face(cg, _p(self.rvp), C.c_int(1), C.c_int(g.Ny))

    def apply_open_boundary_relaxation(self):
        """apply_open_boundary_relaxation! (acoustic_substepping.jl:1339-1361)"""
        g, cg, op, a = self.grid, C.byref(self.cg), self.lateral_open, C.c_double(self.open_boundary_relaxation)
        args = (_p(self.rp), _p(self.rthp), _p(self.rho_d), _p(self.rtheta))
        if g.topo[0] == BOUNDED and op["west"]:
            self.lib.og_relax_open_boundary(cg, *args, C.c_int(0), C.c_int(0), C.c_int(-1), a)
        if g.topo[0] == BOUNDED and op["east"]:
            self.lib.og_relax_open_boundary(cg, *args, C.c_int(0), C.c_int(g.Nx - 1), C.c_int(g.Nx), a)
        if g.topo[1] == BOUNDED and op["south"]:
            self.lib.og_relax_open_boundary(cg, *args, C.c_int(1), C.c_int(0), C.c_int(-1), a)
        if g.topo[1] == BOUNDED and op["north"]:
            self.lib.og_relax_open_boundary(cg, *args, C.c_int(1), C.c_int(g.Ny - 1), C.c_int(g.Ny), a)

    def _halo_w(self, f):
        self.lib.og_fill_halo_periodic_xy(C.byref(self.cg), _p(f), C.c_int(f.shape[0]))
        self.lib.og_fill_halo_z_wall(C.byref(self.cg), _p(f))

    def _eval(self, value, loc):
        g = self.grid
        shape = (g.Nz + (1 if loc[2] == "f" else 0), g.Ny, g.Nx)
        if callable(value):
            x, y, z = g.nodes(loc)
            value = value(x, y, z)
        value = np.asarray(value, dtype=np.float64)
        if value.ndim == 1:        # a column (e.g. ref.density interior)
            value = value[:, None, None]
        return np.broadcast_to(value, shape)

    # -- set! (set_atmosphere_model.jl:198-362, compressible_time_stepping.jl:105-150) ------------
    def set(self, rho=None, theta=None, u=None, v=None, w=None, qv=None, qcl=None, qr=None):
        g = self.grid
        I = g.interior
        if rho is not None:
            I(self.rho_d)[...] = self._eval(rho, "ccc")
            self._halo_center(self.rho_d)
        if qv is not None:
            I(self.q)[...] = self._eval(qv, "ccc")
            I(self.rq)[...] = I(self.rho_d) * I(self.q)
        condensate = 0.0
        if self.microphysics == "Kessler":                    # settable specific microphysical names: rho*q only
            if qcl is not None:
                I(self.rqcl)[...] = I(self.rho_d) * self._eval(qcl, "ccc")
            if qr is not None:
                I(self.rqr)[...] = I(self.rho_d) * self._eval(qr, "ccc")
            condensate = I(self.rqcl) + (I(self.rqr) + 0.0)
        if rho is not None:                                   # total density given: rho_d = rho - (rho q + condensates)
            I(self.rho)[...] = I(self.rho_d)
            I(self.rho_d)[...] = I(self.rho) - (I(self.rq) + condensate)
            self._halo_center(self.rho)
            self._halo_center(self.rho_d)
        if theta is not None:
            I(self.theta)[...] = self._eval(theta, "ccc")
            I(self.rtheta)[...] = I(self.rho_d) * I(self.theta)
        r = self.rho_d
        Hx, Hy, Hz, Nx, Ny, Nz = g.Hx, g.Hy, g.Hz, g.Nx, g.Ny, g.Nz
        if u is not None:
            I(self.u)[...] = self._eval(u, "fcc")
            rx = I(r) if g.topo[0] == FLAT else (I(r) + r[Hz:Hz + Nz, Hy:Hy + Ny, Hx - 1:Hx - 1 + Nx]) / 2
            I(self.ru)[...] = rx * I(self.u)
        if v is not None:
            I(self.v)[...] = self._eval(v, "cfc")
            ry = I(r) if g.topo[1] == FLAT else (I(r) + r[Hz:Hz + Nz, Hy - 1:Hy - 1 + Ny, Hx:Hx + Nx]) / 2
            I(self.rv)[...] = ry * I(self.v)
        if w is not None:
            I(self.w, True)[...] = self._eval(w, "ccf")
            rz = (r[Hz:Hz + Nz + 1, Hy:Hy + Ny, Hx:Hx + Nx] + r[Hz - 1:Hz + Nz, Hy:Hy + Ny, Hx:Hx + Nx]) / 2
            I(self.rw, True)[...] = rz * I(self.w, True)
        self.update_state(compute_tendencies=False)

    # -- update_state! ---------------------------------------------------------
    def update_state(self, compute_tendencies=True):
        cg, L = C.byref(self.cg), self.lib
        kes = self.microphysics == "Kessler"
        if kes:
            g = self.grid
            g.interior(self.rho)[...] = g.interior(self.rho_d) + (g.interior(self.rq) + (g.interior(self.rqcl) + (g.interior(self.rqr) + 0.0)))
        else:
            L.og_total_density(cg, _p(self.rho), _p(self.rho_d), _p(self.rq))
        self._halo_center(self.rho)
        for f in (self.rho_d, self.ru, self.rv, self.rtheta, self.rq) + ((self.rqcl, self.rqr) if kes else ()):
            self._halo_center(f)
        self._halo_w(self.rw)
        L.og_compute_velocities_3d(cg, _p(self.u), _p(self.v), _p(self.w), _p(self.ru), _p(self.rv),
                                   _p(self.rw), _p(self.rho_d))
        self._halo_center(self.u)
        self._halo_center(self.v)
        self._halo_w(self.w)
        if kes:
            self._kessler_thermo()
        elif self.microphysics == "SaturationAdjustment":
            self._sa_thermo()
        else:
            L.og_compressible_thermo(cg, _p(self.theta), _p(self.q), _p(self.T), _p(self.p), _p(self.rho_d),
                                     _p(self.rho), _p(self.rtheta), _p(self.rq),
                                     C.c_double(self.newton[0]), C.c_int(self.newton[1]))
        extra = (self.qcl, self.qr) if kes else ((self.qv, self.ql) if self.microphysics == "SaturationAdjustment" else ())
        for f in (self.theta, self.q, self.T, self.p) + extra:
            self._halo_center(f)
        if compute_tendencies:
            # moisture: total density carrier, acoustic-mean transport velocities
            # (update_atmosphere_model_state.jl:330-343, acoustic_runge_kutta_3.jl:352-358);
            # the momentum / theta / rho_d tendencies computed here by the reference are overwritten
            # by compute_slow_*_tendencies! before they are used.
            L.og_set_weno_order(C.c_int(self.weno_order))
            L.og_set_weno_ft2(C.c_int(getattr(self, "weno_ft2", 0)))      # the FT2 hypothesis (oracle.py: OracleModel.weno_ft2)
            L.og_scalar_tendency_3d(cg, _p(self.G["rq"]), _p(self.rho), _p(self.au), _p(self.av), _p(self.aw), _p(self.q))
            if kes:
                L.og_scalar_tendency_3d(cg, _p(self.G["rqcl"]), _p(self.rho), _p(self.au), _p(self.av), _p(self.aw), _p(self.qcl))
                L.og_scalar_tendency_3d(cg, _p(self.G["rqr"]), _p(self.rho), _p(self.au), _p(self.av), _p(self.aw), _p(self.qr))
            L.og_set_weno_order(C.c_int(5))
            L.og_set_weno_ft2(C.c_int(0))

    def _sa_thermo(self):
        """maybe_adjust_thermodynamic_state on the LiquidIceDensityState of every cell (theta = rho theta / rho_d, q^t = rho q / rho,
        total rho), update_microphysical_fields!, then _compute_temperature_and_pressure! with the adjusted fractions — the same
        Newton inversion, so one temperature (compressible_time_stepping.jl:191-250; saturation_adjustment.jl:236-301)."""
        from .thermo import adjust_warm_phase_density, mixture_gas_constant
        g, tc = self.grid, self.thermo
        I = g.interior
        rd, r = I(self.rho_d), I(self.rho)
        th, qt = I(self.rtheta) / rd, I(self.rq) / r
        I(self.theta)[...], I(self.q)[...] = th, qt
        T, qv, ql, p = I(self.T), I(self.qv), I(self.ql), I(self.p)
        for idx in np.ndindex(th.shape):
            T[idx], qv[idx], ql[idx] = adjust_warm_phase_density(float(th[idx]), float(qt[idx]), float(r[idx]), self.pst, tc,
                                                                 self.sa_solver[0], self.sa_solver[1], self.newton[0], self.newton[1])
            p[idx] = r[idx] * mixture_gas_constant(qv[idx], ql[idx], 0.0, tc) * T[idx]

    def _kessler_thermo(self):
        """theta, q^v, q^cl, q^r (update_microphysical_auxiliaries!) then _compute_temperature_and_pressure! with the fresh
        moisture fractions (q^v, q^cl + q^r): LiquidIceDensityState Newton inversion with the latent term
        (compressible_time_stepping.jl:191-242; dynamic_states.jl:197-232)."""
        g, c, t = self.grid, self.constants, self.tetens
        I = g.interior
        rd, r = I(self.rho_d), I(self.rho)
        th, qv = I(self.rtheta) / rd, I(self.rq) / r
        qcl, qr = I(self.rqcl) / r, I(self.rqr) / r
        I(self.theta)[...], I(self.q)[...], I(self.qcl)[...], I(self.qr)[...] = th, qv, qcl, qr
        ql = qcl + qr
        qd = 1.0 - (qv + ql + 0.0)
        Rm = qd * c.Rd + qv * c.Rv
        cpm = qd * c.cpd + qv * c.cpv + ql * t.cl + 0.0
        kap, gam = Rm / cpm, cpm / (cpm - Rm)
        Lt = (t.Ll * ql + 0.0) / cpm
        T = th ** gam * (r * Rm / self.pst) ** (gam - 1.0) + Lt
        dT = T.copy()
        abstol, maxiter = self.newton
        it = np.zeros(T.shape, dtype=int)
        while True:
            active = (np.abs(dT) > max(abstol, 0.0)) & (it < maxiter)
            if not active.any():
                break
            Phi = (r * Rm * T / self.pst) ** kap * th
            step = -(T - Phi - Lt) / (1.0 - kap * Phi / T)
            dT = np.where(active, step, dT)
            T = np.where(active, T + step, T)
            it = it + active
        I(self.T)[...] = T
        I(self.p)[...] = r * Rm * T
    # -- acoustic stage ----------------------------------------------------------
    def refresh_linearization(self):
        self.lib.og_linearization(C.byref(self.cg), _p(self.Pi), _p(self.thL), _p(self.gR), _p(self.p),
                                  _p(self.rho_d), _p(self.rtheta), _p(self.rho), _p(self.q))
        if self.microphysics == "Kessler":      # gamma R_m with the liquid fraction (acoustic_substepping.jl:376-399)
            g, c, t = self.grid, self.constants, self.tetens
            I = g.interior
            qv, ql = I(self.q), I(self.qcl) + I(self.qr)
            qd = 1 - qv - ql - 0.0
            Rm = qd * c.Rd + qv * c.Rv
            cpm = qd * c.cpd + qv * c.cpv + ql * t.cl + 0.0 * 0.0
            I(self.gR)[...] = cpm * Rm / (cpm - Rm)
        elif self.microphysics == "SaturationAdjustment":
            g, c, tc = self.grid, self.constants, self.thermo
            I = g.interior
            qv, ql = I(self.qv), I(self.ql)
            qd = 1 - qv - ql - 0.0
            Rm = qd * c.Rd + qv * c.Rv
            cpm = qd * c.cpd + qv * c.cpv + ql * tc.cl + 0.0 * 0.0
            I(self.gR)[...] = cpm * Rm / (cpm - Rm)
        for f in (self.Pi, self.thL, self.gR):
            self._halo_sub(f)       # fill_halo_regions! with the default conditions (acoustic_substepping.jl:365-367): zero gradient on Bounded x / y

    def seed_time_averaged_velocities(self):
        self.au[...] = self.u
        self.av[...] = self.v
        self.aw[...] = self.w

    def compute_slow_tendencies(self):
        cg, L, G = C.byref(self.cg), self.lib, self.G
        L.og_set_weno_order(C.c_int(self.weno_order))           # process-wide switch of the C library (oracle.py)
        L.og_set_weno_ft2(C.c_int(getattr(self, "weno_ft2", 0)))
        L.og_u_tendency(cg, _p(G["ru"]), _p(self.ru), _p(self.rv), _p(self.rw), _p(self.u))
        L.og_v_tendency(cg, _p(G["rv"]), _p(self.ru), _p(self.rv), _p(self.rw), _p(self.v))
        L.og_w_tendency_slow(cg, _p(G["rw"]), _p(self.ru), _p(self.rv), _p(self.rw), _p(self.w))
        L.og_density_tendency(cg, _p(G["rho_d"]), _p(self.ru), _p(self.rv), _p(self.rw))
        L.og_scalar_tendency_3d(cg, _p(G["rtheta"]), _p(self.rho_d), _p(self.u), _p(self.v), _p(self.w), _p(self.theta))
        L.og_set_weno_order(C.c_int(5))
        L.og_set_weno_ft2(C.c_int(0))
        if self.coriolis_f != 0.0:      # - x_f_cross_U, - y_f_cross_U of an FPlane (dynamics_kernel_functions.jl:79,99)
            from .forcings import _xy_to_cf, _xy_to_fc
            g = self.grid
            g.interior(G["ru"])[...] -= -self.coriolis_f * _xy_to_fc(self, self.rv)
            g.interior(G["rv"])[...] -= self.coriolis_f * _xy_to_cf(self, self.ru)
        if self.relaxation:
            from .forcings import add_relaxation_tendencies
            add_relaxation_tendencies(self)
        if getattr(self, "field_forcing", None) is not None:
            from .forcings import add_field_forcing
            add_field_forcing(self)

    def assemble_slow_vertical_momentum(self):
        pr = _p(self.ref.pressure) if self.ref is not None else None
        rr = _p(self.ref.density) if self.ref is not None else None
        self.lib.og_slow_vertical_momentum(C.byref(self.cg), _p(self.Gs), _p(self.G["rw"]), _p(self.p),
                                           _p(self.rho), pr, rr)

    def implicit_damping_factors(self):
        td, g = self.td, self.grid
        if td.damping_coefficient is None or not td.damp_vertical or getattr(td, "direct_damping", False):
            return 0.0, 0.0
        base = td.damping_coefficient * float(np.min(g.dzc[g.Hz:g.Hz + g.Nz])) ** 2
        return td.forward_weight * base, (1 - td.forward_weight) * base

    def acoustic_substep_loop(self, dt, beta):
        """acoustic_rk3_substep_loop! (acoustic_substepping.jl:1404-1590)"""
        g, td, c = self.grid, self.td, self.constants
        cg, L = C.byref(self.cg), self.lib
        Nz = g.Nz
        n_tau, dtau = stage_substep_count_and_size(td.substeps, beta, float(dt), g, c, td.acoustic_cfl, td.substep_distribution)
        self.last_substeps.append(n_tau)
        om = td.forward_weight
        dtn, dto = om * dtau, (1 - om) * dtau
        self.assemble_slow_vertical_momentum()
        # initialize_stage_perturbations!
        for f in (self.rth_old, self.rs, self.rths):
            g.interior(f)[...] = 0.0
        g.interior(self.au)[...] = 0.0
        g.interior(self.av)[...] = 0.0
        g.interior(self.aw)[...] = 0.0          # :xyz launch: faces 0..Nz-1
        for prime, name, nk in ((self.rp, "rho_d", Nz), (self.rthp, "rtheta", Nz), (self.rup, "ru", Nz),
                                (self.rvp, "rv", Nz), (self.rwp, "rw", Nz)):
            L.og_initialize_perturbation(cg, _p(prime), _p(self.U0[name]), _p(getattr(self, name)), C.c_int(nk))
        for f in (self.rp, self.rthp):
            self._halo_sub(f)
        self._halo_sub(self.rup, xface=True)
        self._halo_sub(self.rvp, yface=True)
        self._halo_w(self.rwp)
        d_new, d_old = self.implicit_damping_factors()
        sponge = None
        if getattr(td, "sponge", None) is not None:
            zf = np.asarray(g.zf, dtype=np.float64)
            self._sponge = upper_sponge_profile(zf, zf[-1] - zf[0], *td.sponge)
            sponge = _p(self._sponge)
        for s in range(1, n_tau + 1):
            gate = apply_horizontal_pressure_gradient_substep(s, n_tau, td.apply_first)
            L.og_explicit_horizontal_step(cg, _p(self.rup), _p(self.rvp), _p(self.p), _p(self.rthp), _p(self.Pi),
                                          _p(self.gR), _p(self.G["ru"]), _p(self.G["rv"]), C.c_double(dtau),
                                          C.c_int(int(gate)))
            self._halo_sub(self.rup, xface=True)
            self._halo_sub(self.rvp, yface=True)
            self.enforce_wall_impenetrability()
            L.og_build_predictors(cg, _p(self.rs), _p(self.rths), _p(self.rth_old), _p(self.rp), _p(self.rthp),
                                  _p(self.rwp), _p(self.rup), _p(self.rvp), _p(self.G["rho_d"]),
                                  _p(self.G["rtheta"]), _p(self.thL), C.c_double(dtau), C.c_double(dto),
                                  C.c_double(td.f_theta))
            self._halo_sub(self.rth_old)
            L.og_build_vertical_rhs(cg, _p(self.rhs), _p(self.rs), _p(self.rths), _p(self.rp), _p(self.rthp),
                                    _p(self.rwp), _p(self.Pi), _p(self.gR), _p(self.Gs), C.c_double(dtau),
                                    C.c_double(dtn), C.c_double(dto), C.c_double(d_old), C.c_double(td.f_w), sponge)
            L.og_acoustic_tridiagonal_solve(cg, _p(self.rwp), _p(self.rhs), _p(self.Pi), _p(self.thL),
                                            _p(self.gR), C.c_double(dtn), C.c_double(d_new), sponge)
            L.og_post_solve_recovery(cg, _p(self.rp), _p(self.rthp), _p(self.rwp), _p(self.rup), _p(self.rvp),
                                     _p(self.rs), _p(self.rths), _p(self.au), _p(self.av), _p(self.aw),
                                     _p(self.thL), C.c_double(dtn))
            self.apply_open_boundary_relaxation()
            self._halo_sub(self.rp)
            self._halo_sub(self.rthp)
            if td.damping_coefficient is not None and getattr(td, "direct_damping", False):
                # DirectDivergenceDamping: delta lives in the density predictor, free between recovery and the next build
                L.og_direct_divergence_damping(cg, _p(self.rup), _p(self.rvp), _p(self.rs), _p(self.thL),
                                               C.c_double(td.damping_coefficient))
            elif td.damping_coefficient is not None:
                L.og_thermal_divergence_damping(cg, _p(self.rup), _p(self.rvp), _p(self.rthp), _p(self.rth_old),
                                                _p(self.thL), C.c_double(td.damping_coefficient), C.c_double(dtau),
                                                C.c_double(getattr(td, "damping_length_scale", None) or 0.0))
            self._halo_sub(self.rup, xface=True)
            self._halo_sub(self.rvp, yface=True)
            self.enforce_wall_impenetrability()
        L.og_finalize_time_averaged_velocity(cg, _p(self.au), _p(self.av), _p(self.aw), _p(self.ru), _p(self.rv),
                                             _p(self.rw), _p(self.rho_d), C.c_double(1.0 / float(n_tau)))
        self._halo_center(self.au)
        self._halo_center(self.av)
        self._halo_w(self.aw)
        L.og_recover_full_state(cg, _p(self.rho_d), _p(self.rtheta), _p(self.ru), _p(self.rv), _p(self.rw),
                                _p(self.rp), _p(self.rthp), _p(self.rup), _p(self.rvp), _p(self.rwp))
        if self._walls():      # the fills with the model's boundary conditions and compute_velocities! are the caller's (as for the library)
            return
        for f in (self.rho_d, self.rtheta, self.ru, self.rv):
            self._halo_center(f)
        self._halo_w(self.rw)
        L.og_compute_velocities_3d(cg, _p(self.u), _p(self.v), _p(self.w), _p(self.ru), _p(self.rv),
                                   _p(self.rw), _p(self.rho_d))
        self._halo_center(self.u)
        self._halo_center(self.v)
        self._halo_w(self.w)

    def acoustic_rk3_substep(self, dt, beta):
        self.refresh_linearization()
        self.compute_slow_tendencies()
        self.acoustic_substep_loop(dt, beta)
        for n in ("rq",) + (("rqcl", "rqr") if self.microphysics == "Kessler" else ()):
            self.lib.og_ws_rk3_scalar(C.byref(self.cg), _p(getattr(self, n)), _p(self.U0[n]), _p(self.G[n]), C.c_double(beta * dt))

    def time_step(self, dt):
        dt = float(dt)
        if self.iteration == 0:                     # maybe_prepare_first_time_step!
            self.seed_time_averaged_velocities()
            self.update_state(compute_tendencies=True)
        for n in self.PROGNOSTIC:                   # store_initial_state!
            self.U0[n][...] = getattr(self, n)
        self.refresh_linearization()                # freeze_linearization_state!
        self.seed_time_averaged_velocities()
        self.last_substeps = []
        for beta in self.BETAS:
            self.acoustic_rk3_substep(dt, beta)
            self.update_state(compute_tendencies=True)
        if self.microphysics == "Kessler":
            self.microphysics_model_update(dt)
        self.clock_time += dt
        self.iteration += 1

    def microphysics_model_update(self, dt):
        """microphysics_model_update!(::DCMIP2016KesslerMicrophysics, model) for CompressibleDynamics: the column kernel with
        density = dynamics_density = rho_d and pressure = dynamics.pressure (dcmip2016_kessler.jl:460-485), then update_state!."""
        from .kessler import kessler_column_update
        g = self.grid
        I = g.interior
        for j in range(g.Ny):
            for i in range(g.Nx):
                rho = np.ascontiguousarray(I(self.rho_d)[:, j, i])
                p = np.ascontiguousarray(I(self.p)[:, j, i])
                cols = [np.ascontiguousarray(I(f)[:, j, i]) for f in (self.theta, self.rtheta, self.rq, self.rqcl, self.rqr)]
                qv, qcl, qr, W, P, _ = kessler_column_update(dt, rho, p, self.pst, g.zc, *cols, self.kessler, self.tetens)
                for f, col in zip((self.theta, self.rtheta, self.rq, self.rqcl, self.rqr), cols):
                    I(f)[:, j, i] = col
                I(self.q)[:, j, i], I(self.qcl)[:, j, i], I(self.qr)[:, j, i], I(self.W)[:, j, i] = qv, qcl, qr, W
                self.precipitation_rate[j, i] = P
        self.update_state(compute_tendencies=True)
