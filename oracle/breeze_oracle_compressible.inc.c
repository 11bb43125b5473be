/*
 * breeze_oracle_compressible.inc.c — CPU restatement of Breeze.jl's compressible
 * split-explicit path.  #included at the end of breeze_oracle.c.
 *
 * TEST INFRASTRUCTURE ONLY (same rules as breeze_oracle.c).
 *
 * PARITY STATUS: the Breeze-side arithmetic below follows the cited reference lines
 * statement by statement; it is pinned by the reference's own known-answer tests restated in
 * tests/test_oracle_compressible.py (closed-form tridiagonal coefficients, explicit horizontal
 * step KAT, first-small-step PGF gate truth table, substep counts, rest-state / structural
 * invariants).  The WENO reconstruction used by the slow tendencies stays "parity unpinned"
 * (breeze_oracle.c header).  Oceananigans operators are recalled: d/dx = delta * (1/Delta),
 * V^-1 = 1/(dx dy dz), Iz(c)[k] = (c[k] + c[k-1])/2.
 *
 * Index convention: 0-based; cell k has faces k (below) and k+1 (above); Julia face K = k+1.
 * Scope: (Periodic|Flat, Periodic|Flat, Bounded), microphysics = nothing (vapour only),
 * ProportionalSubsteps, ThermalDivergenceDamping (optionally damp_vertical), DirectDivergenceDamping or none, optional UpperSponge
 * (a per-face profile rate * ramp(z) built by the caller: acoustic_substepping.jl:584-602).
 */

/* boundary-aware centre->face z interpolation (acoustic_substepping.jl:539-550) */
static inline double ibz_face(const og_grid *G, const double *psi, size_t n /* index of cell k */, int k)
{
    double fp = psi[n], fm = psi[n - STRZ(G)];
    int pp = (k < 0) || (k >= G->Nz);          /* cell k peripheral   */
    int pm = (k - 1 < 0) || (k - 1 >= G->Nz);  /* cell k-1 peripheral */
    fp = pp ? fm : fp;
    fm = pm ? fp : fm;
    return (fp + fm) / 2.0;
}

static inline double rdzc_at(const og_grid *G, int k) { return 1.0 / G->dzc[k + G->Hz]; }
static inline double rdzf_at(const og_grid *G, int k) { return 1.0 / G->dzf[k + G->Hz]; }

/* ------------------------------------------------------------------------- */
/* total density rho = rho_d + rho q  (microphysics_interface.jl:635-659,     */
/* compressible_time_stepping.jl:83-103)                                      */
/* ------------------------------------------------------------------------- */
void og_total_density(const og_grid *G, double *rho, const double *rho_d, const double *rq)
{
#pragma omp parallel for collapse(2) schedule(static)
    for (int k = 0; k < G->Nz; ++k)
        for (int j = 0; j < G->Ny; ++j)
            for (int i = 0; i < G->Nx; ++i) {
                size_t n = IDX(G, i, j, k);
                rho[n] = rho_d[n] + (rq[n] + 0.0);
            }
}

/* velocities with a 3-D carrier density (update_atmosphere_model_state.jl:248-254) */
void og_compute_velocities_3d(const og_grid *G, double *u, double *v, double *w,
                              const double *ru, const double *rv, const double *rw, const double *rho)
{
#pragma omp parallel for collapse(2) schedule(static)
    for (int k = 0; k <= G->Nz; ++k)
        for (int j = 0; j < G->Ny; ++j)
            for (int i = 0; i < G->Nx; ++i) {
                size_t n = IDX(G, i, j, k);
                double rx = (G->tx == FLAT) ? rho[n] : (rho[n] + rho[n - 1]) / 2.0;
                double ry = (G->ty == FLAT) ? rho[n] : (rho[n] + rho[n - STRY(G)]) / 2.0;
                double rz = (rho[n] + rho[n - STRZ(G)]) / 2.0;
                if (k < G->Nz) {
                    u[n] = ru[n] / rx;
                    v[n] = rv[n] / ry;
                }
                w[n] = rw[n] / rz;
            }
}

/* ------------------------------------------------------------------------- */
/* a17: theta, qv, T, p  (update_atmosphere_model_state.jl:256-292,           */
/* potential_temperature_formulation.jl:115-123, compressible_time_stepping.jl*/
/* :191-242, dynamic_states.jl:197-232, Solvers.jl NewtonSolver abstol 1e-4,   */
/* reltol 0, maxiter 8)                                                       */
/* ------------------------------------------------------------------------- */
void og_compressible_thermo(const og_grid *G, double *theta, double *qv, double *T, double *p,
                            const double *rho_d, const double *rho, const double *rtheta,
                            const double *rq, double abstol, int maxiter)
{
#pragma omp parallel for collapse(2) schedule(static)
    for (int k = 0; k < G->Nz; ++k)
        for (int j = 0; j < G->Ny; ++j)
            for (int i = 0; i < G->Nx; ++i) {
                size_t n = IDX(G, i, j, k);
                double th = rtheta[n] / rho_d[n];
                double r = rho[n];
                double q = rq[n] / r;
                theta[n] = th;
                qv[n] = q;
                double qd = 1.0 - (q + 0.0 + 0.0);
                double Rm = qd * G->Rd + q * G->Rv;
                double cpm = qd * G->cpd + q * G->cpv + 0.0 + 0.0;
                double kap = Rm / cpm;
                double gam = cpm / (cpm - Rm);
                double L = 0.0;
                double Tn = pow(th, gam) * pow(r * Rm / G->p_st, gam - 1.0) + L;
                double dT = Tn;
                int it = 0;
                while (fabs(dT) > fmax(abstol, 0.0 * Tn) && it < maxiter) {
                    double Phi = pow(r * Rm * Tn / G->p_st, kap) * th;
                    dT = -(Tn - Phi - L) / (1.0 - kap * Phi / Tn);
                    Tn += dT;
                    ++it;
                }
                T[n] = Tn;
                p[n] = r * Rm * Tn;
            }
}

/* ------------------------------------------------------------------------- */
/* a15: stage-entry linearisation (acoustic_substepping.jl:340-399)           */
/* ------------------------------------------------------------------------- */
void og_linearization(const og_grid *G, double *Pi, double *thL, double *gR,
                      const double *p, const double *rho_d, const double *rtheta,
                      const double *rho, const double *qv)
{
    double kap = G->Rd / G->cpd;
#pragma omp parallel for collapse(2) schedule(static)
    for (int k = 0; k < G->Nz; ++k)
        for (int j = 0; j < G->Ny; ++j)
            for (int i = 0; i < G->Nx; ++i) {
                size_t n = IDX(G, i, j, k);
                Pi[n] = pow(p[n] / G->p_st, kap);
                double rh = (rho_d[n] == 0.0) ? 1.0 : rho_d[n];
                thL[n] = rtheta[n] / rh;
                double q = qv[n];
                double qd = 1.0 - q - 0.0 - 0.0;
                double Rm = qd * G->Rd + q * G->Rv;
                double cpm = qd * G->cpd + q * G->cpv + 0.0 * 0.0 + 0.0 * 0.0;
                double cvm = cpm - Rm;
                gR[n] = cpm * Rm / cvm;
                (void)rho;
            }
}

/* ------------------------------------------------------------------------- */
/* slow tendencies                                                            */
/* ------------------------------------------------------------------------- */

/* G_rho_d = -div(rho u)  (compressible_density_tendency.jl:52-55) */
void og_density_tendency(const og_grid *G, double *Gr, const double *ru, const double *rv, const double *rw)
{
#pragma omp parallel for collapse(2) schedule(static)
    for (int k = 0; k < G->Nz; ++k)
        for (int j = 0; j < G->Ny; ++j)
            for (int i = 0; i < G->Nx; ++i) {
                double dz = dzc_at(G, k);
                double Ax = G->dy * dz, Ay = G->dx * dz, Az = G->dx * G->dy;
                double Vinv = 1.0 / (G->dx * G->dy * dz);
                size_t n = IDX(G, i, j, k);
                double a = 0.0, b = 0.0, c = 0.0;
                if (G->tx != FLAT) a = Ax * ru[n + 1] - Ax * ru[n];
                if (G->ty != FLAT) b = Ay * rv[n + STRY(G)] - Ay * rv[n];
                c = Az * rw[n + STRZ(G)] - Az * rw[n];
                Gr[n] = -(Vinv * (a + b + c));
            }
}

/* scalar tendency with a 3-D carrier density (src/Advection.jl:20-35) */
static inline double flux_x_scalar3(const og_grid *G, const double *rho, const double *u, const double *c, int i, int j, int k)
{
    size_t n = IDX(G, i, j, k);
    double ut = u[n];
    double cr = biased_face(c + n, 1, left_bias(ut), i, G->Nx, G->tx == BOUNDED);
    double Ax = G->dy * G->dzc[k + G->Hz];
    return ((rho[n] + rho[n - 1]) / 2.0) * (Ax * ut * cr);
}
static inline double flux_y_scalar3(const og_grid *G, const double *rho, const double *v, const double *c, int i, int j, int k)
{
    size_t n = IDX(G, i, j, k);
    double vt = v[n];
    double cr = biased_face(c + n, (ptrdiff_t)SX(G), left_bias(vt), j, G->Ny, G->ty == BOUNDED);
    double Ay = G->dx * G->dzc[k + G->Hz];
    return ((rho[n] + rho[n - STRY(G)]) / 2.0) * (Ay * vt * cr);
}
static inline double flux_z_scalar3(const og_grid *G, const double *rho, const double *w, const double *c, int i, int j, int k)
{
    size_t n = IDX(G, i, j, k);
    double wt = w[n];
    double cr = biased_face(c + n, (ptrdiff_t)(SX(G) * SY(G)), left_bias(wt), k, G->Nz, G->tz == BOUNDED);
    double Az = G->dx * G->dy;
    return ((rho[n] + rho[n - STRZ(G)]) / 2.0) * (Az * wt * cr);
}

void og_scalar_tendency_3d(const og_grid *G, double *Gc, const double *rho, const double *u,
                           const double *v, const double *w, const double *c)
{
#pragma omp parallel for collapse(2) schedule(static)
    for (int k = 0; k < G->Nz; ++k)
        for (int j = 0; j < G->Ny; ++j)
            for (int i = 0; i < G->Nx; ++i) {
                double Vinv = 1.0 / (G->dx * G->dy * G->dzc[k + G->Hz]);
                double dxF = 0.0, dyF = 0.0, dzF = 0.0;
                if (G->tx != FLAT)
                    dxF = flux_x_scalar3(G, rho, u, c, i + 1, j, k) - flux_x_scalar3(G, rho, u, c, i, j, k);
                if (G->ty != FLAT)
                    dyF = flux_y_scalar3(G, rho, v, c, i, j + 1, k) - flux_y_scalar3(G, rho, v, c, i, j, k);
                dzF = flux_z_scalar3(G, rho, w, c, i, j, k + 1) - flux_z_scalar3(G, rho, w, c, i, j, k);
                Gc[IDX(G, i, j, k)] = -(Vinv * (dxF + dyF + dzF));
            }
}

/* slow z-momentum tendency: advection only (SlowTendencyMode, dynamics_interface.jl:397-411) */
void og_w_tendency_slow(const og_grid *G, double *Gw, const double *ru, const double *rv,
                        const double *rw, const double *w)
{
#pragma omp parallel for collapse(2) schedule(static)
    for (int k = 1; k < G->Nz; ++k)
        for (int j = 0; j < G->Ny; ++j)
            for (int i = 0; i < G->Nx; ++i) {
                double Vinv = 1.0 / (G->dx * G->dy * G->dzf[k + G->Hz]);
                double a = 0.0, b = 0.0, c = 0.0;
                if (G->tx != FLAT) a = F_Uw(G, ru, w, i + 1, j, k) - F_Uw(G, ru, w, i, j, k);
                if (G->ty != FLAT) b = F_Vw(G, rv, w, i, j + 1, k) - F_Vw(G, rv, w, i, j, k);
                c = F_Ww(G, rw, w, i, j, k) - F_Ww(G, rw, w, i, j, k - 1);
                Gw[IDX(G, i, j, k)] = -(Vinv * (a + b + c));
            }
}

/* G^s_rho_w (acoustic_substepping.jl:727-752).  p_ref / rho_ref are halo-inclusive columns
 * (entry k+Hz) or NULL for reference_state = nothing.  Bottom face is masked to 0. */
void og_slow_vertical_momentum(const og_grid *G, double *Gs, const double *Gw, const double *p,
                               const double *rho, const double *p_ref, const double *rho_ref)
{
#pragma omp parallel for collapse(2) schedule(static)
    for (int k = 0; k < G->Nz; ++k)
        for (int j = 0; j < G->Ny; ++j)
            for (int i = 0; i < G->Nx; ++i) {
                size_t n = IDX(G, i, j, k), m = n - STRZ(G);
                double dp, rf;
                if (p_ref) {
                    const double *pr = p_ref + G->Hz, *rr = rho_ref + G->Hz;
                    dp = ((p[n] - pr[k]) - (p[m] - pr[k - 1])) * rdzf_at(G, k);
                    rf = ((rho[n] - rr[k]) + (rho[m] - rr[k - 1])) / 2.0;
                } else {
                    dp = (p[n] - p[m]) * rdzf_at(G, k);
                    rf = (rho[n] + rho[m]) / 2.0;
                }
                Gs[n] = (Gw[n] - dp - G->g * rf) * (k > 0 ? 1.0 : 0.0);
            }
}

/* ------------------------------------------------------------------------- */
/* substep kernels                                                            */
/* ------------------------------------------------------------------------- */

/* _initialize_stage_perturbations! + rewind of rho w  (acoustic_substepping.jl:815-838) */
void og_initialize_perturbation(const og_grid *G, double *prime, const double *outer, const double *stage, int nk)
{
#pragma omp parallel for collapse(2) schedule(static)
    for (int k = 0; k < nk; ++k)
        for (int j = 0; j < G->Ny; ++j)
            for (int i = 0; i < G->Nx; ++i) {
                size_t n = IDX(G, i, j, k);
                prime[n] = outer[n] - stage[n];
            }
}

/* _explicit_horizontal_step! (acoustic_substepping.jl:860-881) */
void og_explicit_horizontal_step(const og_grid *G, double *rup, double *rvp, const double *p,
                                 const double *rthp, const double *Pi, const double *gR,
                                 const double *Gu, const double *Gv, double dtau, int apply_pgf)
{
    double f = apply_pgf ? 1.0 : 0.0;
#pragma omp parallel for collapse(2) schedule(static)
    for (int k = 0; k < G->Nz; ++k)
        for (int j = 0; j < G->Ny; ++j)
            for (int i = 0; i < G->Nx; ++i) {
                size_t n = IDX(G, i, j, k);
                if (G->tx != FLAT) {
                    size_t m = n - 1;
                    double dpL = (p[n] - p[m]) * (1.0 / G->dx);
                    double dpp = (gR[n] * Pi[n] * rthp[n] - gR[m] * Pi[m] * rthp[m]) * (1.0 / G->dx);
                    double dp = dpL + f * dpp;
                    rup[n] += dtau * (Gu[n] - dp);
                } else
                    rup[n] += dtau * Gu[n];      /* the kernel updates both components; a Flat direction has a zero gradient */
                if (G->ty != FLAT) {
                    size_t m = n - STRY(G);
                    double dpL = (p[n] - p[m]) * (1.0 / G->dy);
                    double dpp = (gR[n] * Pi[n] * rthp[n] - gR[m] * Pi[m] * rthp[m]) * (1.0 / G->dy);
                    double dp = dpL + f * dpp;
                    rvp[n] += dtau * (Gv[n] - dp);
                } else
                    rvp[n] += dtau * Gv[n];
            }
}

/* _build_predictors! (acoustic_substepping.jl:907-927) */
void og_build_predictors(const og_grid *G, double *rs, double *rths, double *rth_old,
                         const double *rp, const double *rthp, const double *rwp,
                         const double *rup, const double *rvp, const double *Grho,
                         const double *Grth, const double *thL, double dtau, double dtau_old, double f_theta)
{
#pragma omp parallel for collapse(2) schedule(static)
    for (int k = 0; k < G->Nz; ++k)
        for (int j = 0; j < G->Ny; ++j)
            for (int i = 0; i < G->Nx; ++i) {
                size_t n = IDX(G, i, j, k), up = n + STRZ(G);
                double dz = dzc_at(G, k);
                double Ax = G->dy * dz, Ay = G->dx * dz;
                double Vinv = 1.0 / (G->dx * G->dy * dz);
                rth_old[n] = rthp[n];
                double dxM = 0.0, dyM = 0.0, dxT = 0.0, dyT = 0.0;
                if (G->tx != FLAT) {
                    dxM = Ax * rup[n + 1] - Ax * rup[n];
                    dxT = Ax * ((thL[n + 1] + thL[n]) / 2.0) * rup[n + 1] - Ax * ((thL[n] + thL[n - 1]) / 2.0) * rup[n];
                }
                if (G->ty != FLAT) {
                    size_t s = STRY(G);
                    dyM = Ay * rvp[n + s] - Ay * rvp[n];
                    dyT = Ay * ((thL[n + s] + thL[n]) / 2.0) * rvp[n + s] - Ay * ((thL[n] + thL[n - s]) / 2.0) * rvp[n];
                }
                double divM = Vinv * (dxM + dyM);
                double divT = Vinv * (dxT + dyT);
                double dzW = (rwp[up] - rwp[n]) * rdzc_at(G, k);
                double dzT = (ibz_face(G, thL, up, k + 1) * rwp[up] - ibz_face(G, thL, n, k) * rwp[n]) * rdzc_at(G, k);
                rs[n] = rp[n] + dtau * (Grho[n] - divM) - dtau_old * dzW;
                rths[n] = rthp[n] + dtau * (f_theta * Grth[n] - divT) - dtau_old * dzT;
            }
}

/* _build_vertical_rhs! (acoustic_substepping.jl:933-970), faces k = 0..Nz */
void og_build_vertical_rhs(const og_grid *G, double *rhs, const double *rs, const double *rths,
                           const double *rp, const double *rthp, const double *rwp,
                           const double *Pi, const double *gR, const double *Gs,
                           double dtau, double dtau_new, double dtau_old, double d_old, double f_w,
                           const double *sponge /* rate * ramp per face 0..Nz, or NULL */)
{
#pragma omp parallel for collapse(2) schedule(static)
    for (int k = 0; k <= G->Nz; ++k)
        for (int j = 0; j < G->Ny; ++j)
            for (int i = 0; i < G->Nx; ++i) {
                size_t n = IDX(G, i, j, k), m = n - STRZ(G), up = n + STRZ(G);
                if (k == 0 || k == G->Nz) {
                    rhs[n] = 0.0;
                    continue;
                }
                double Cn = gR[n] * Pi[n], Cm = gR[m] * Pi[m];
                double dps = (Cn * rths[n] - Cm * rths[m]) * rdzf_at(G, k);
                double dpo = (Cn * rthp[n] - Cm * rthp[m]) * rdzf_at(G, k);
                double Gp = dtau_old * dpo + dtau_new * dps;
                double rfs = (rs[n] + rs[m]) / 2.0;
                double rfo = (rp[n] + rp[m]) / 2.0;
                double Gb = G->g * (dtau_old * rfo + dtau_new * rfs);
                double d2 = ((rwp[up] - rwp[n]) * rdzc_at(G, k) - (rwp[n] - rwp[m]) * rdzc_at(G, k - 1)) * rdzf_at(G, k);
                double Gd = -d_old * d2;
                double Gsp = sponge ? fabs(dtau_old) * sponge[k] * rwp[n] : 0.0;      /* sponge_rhs, :598-603 */
                rhs[n] = rwp[n] + dtau * f_w * Gs[n] - Gp - Gb - Gd - Gsp;
            }
}

/* get_coefficient for the acoustic column system (acoustic_substepping.jl:605-659).
 * Row r <-> face r (0-based face r = Julia row r+1).  which: 0 lower (sub-diagonal of row r,
 * i.e. Julia get_coefficient(k = r, Lower), kf = r+1 in Julia = face r here), 1 diagonal, 2 upper. */
static inline double acoustic_coefficient(const og_grid *G, const double *Pi, const double *thL,
                                          const double *gR, int i, int j, int r, int which,
                                          double dtn, double d_new, const double *sponge)
{
    size_t n = IDX(G, i, j, r), m = n - STRZ(G);
    double rdf = rdzf_at(G, r);
    if (which == 0) {
        double rdm = rdzc_at(G, r - 1);
        double Cm = gR[m] * Pi[m];
        double thm = ibz_face(G, thL, m, r - 1);
        double pgf = -(dtn * dtn) * Cm * thm * rdm * rdf;
        double buoy = (dtn * dtn) * G->g * rdm / 2.0;
        double damp = -d_new * rdm * rdf;
        return pgf + buoy + damp;
    } else if (which == 1) {
        double rdp = rdzc_at(G, r), rdm = rdzc_at(G, r - 1);
        double Cp = gR[n] * Pi[n], Cm = gR[m] * Pi[m];
        double th = ibz_face(G, thL, n, r);
        double pgf = (dtn * dtn) * th * (Cp * rdp + Cm * rdm) * rdf;
        double buoy = (dtn * dtn) * G->g * (rdp - rdm) / 2.0;
        double damp = d_new * (rdp + rdm) * rdf;
        double spg = sponge ? fabs(dtn) * sponge[r] : 0.0;                              /* sponge_term_diag, :591-596 */
        return 1.0 + (pgf + buoy + damp + spg) * (r > 0 ? 1.0 : 0.0);
    } else {
        double rdp = rdzc_at(G, r);
        double Cp = gR[n] * Pi[n];
        double thp = ibz_face(G, thL, n + STRZ(G), r + 1);
        double pgf = -(dtn * dtn) * Cp * thp * rdp * rdf;
        double buoy = -(dtn * dtn) * G->g * rdp / 2.0;
        double damp = -d_new * rdp * rdf;
        return (pgf + buoy + damp) * (r > 0 ? 1.0 : 0.0);
    }
}

double og_acoustic_coefficient(const og_grid *G, const double *Pi, const double *thL, const double *gR,
                               int i, int j, int r, int which, double dtn, double d_new)
{
    return acoustic_coefficient(G, Pi, thL, gR, i, j, r, which, dtn, d_new, NULL);
}

/* BatchedTridiagonalSolver (Oceananigans, recalled): Thomas, rows r = 0..Nz-1, result into rwp */
void og_acoustic_tridiagonal_solve(const og_grid *G, double *rwp, const double *rhs, const double *Pi,
                                   const double *thL, const double *gR, double dtn, double d_new, const double *sponge)
{
    const double EPS10 = 10.0 * 2.220446049250313e-16;
    int Nz = G->Nz;
#pragma omp parallel for collapse(2) schedule(static)
    for (int j = 0; j < G->Ny; ++j)
        for (int i = 0; i < G->Nx; ++i) {
            double *t = (double *)malloc(sizeof(double) * (size_t)(Nz + 1));
            double beta = acoustic_coefficient(G, Pi, thL, gR, i, j, 0, 1, dtn, d_new, sponge);
            size_t n0 = IDX(G, i, j, 0);
            rwp[n0] = rhs[n0] / beta;
            for (int r = 1; r < Nz; ++r) {
                size_t n = IDX(G, i, j, r), m = n - STRZ(G);
                double cm = acoustic_coefficient(G, Pi, thL, gR, i, j, r - 1, 2, dtn, d_new, sponge);
                double b = acoustic_coefficient(G, Pi, thL, gR, i, j, r, 1, dtn, d_new, sponge);
                double a = acoustic_coefficient(G, Pi, thL, gR, i, j, r, 0, dtn, d_new, sponge);
                t[r] = cm / beta;
                beta = b - a * t[r];
                if (fabs(beta) > EPS10) rwp[n] = (rhs[n] - a * rwp[m]) / beta;
            }
            for (int r = Nz - 2; r >= 0; --r) {
                size_t n = IDX(G, i, j, r);
                rwp[n] -= t[r + 1] * rwp[n + STRZ(G)];
            }
            free(t);
        }
}

/* _post_solve_recovery! (acoustic_substepping.jl:993-1002) */
void og_post_solve_recovery(const og_grid *G, double *rp, double *rthp, const double *rwp,
                            const double *rup, const double *rvp, const double *rs, const double *rths,
                            double *au, double *av, double *aw, const double *thL, double dtau_new)
{
#pragma omp parallel for collapse(2) schedule(static)
    for (int k = 0; k < G->Nz; ++k)
        for (int j = 0; j < G->Ny; ++j)
            for (int i = 0; i < G->Nx; ++i) {
                size_t n = IDX(G, i, j, k), up = n + STRZ(G);
                double dzW = (rwp[up] - rwp[n]) * rdzc_at(G, k);
                double dzT = (ibz_face(G, thL, up, k + 1) * rwp[up] - ibz_face(G, thL, n, k) * rwp[n]) * rdzc_at(G, k);
                rp[n] = rs[n] - dtau_new * dzW;
                rthp[n] = rths[n] - dtau_new * dzT;
                au[n] += rup[n];
                av[n] += rvp[n];
                aw[n] += rwp[n];
            }
}

/* _thermal_divergence_damping! (acoustic_substepping.jl:1123-1139); kappa = alpha*lmin^2/dtau (LocalHorizontalDampingScale, :1100-1110)
 * or, with ThermalDivergenceDamping(length_scale = l), (alpha l^2)/dtau (FixedHorizontalDampingScale, :1085-1092; length_scale <= 0: none) */
void og_thermal_divergence_damping(const og_grid *G, double *rup, double *rvp, const double *rthp,
                                   const double *rth_old, const double *thL, double alpha, double dtau, double length_scale)
{
    double lmin = INFINITY;
    if (G->tx != FLAT) lmin = fmin(lmin, G->dx);
    if (G->ty != FLAT) lmin = fmin(lmin, G->dy);
    double kap = (length_scale > 0.0) ? (alpha * (length_scale * length_scale)) / dtau : alpha * (lmin * lmin) / dtau;
#pragma omp parallel for collapse(2) schedule(static)
    for (int k = 0; k < G->Nz; ++k)
        for (int j = 0; j < G->Ny; ++j)
            for (int i = 0; i < G->Nx; ++i) {
                size_t n = IDX(G, i, j, k);
                if (G->tx != FLAT) {
                    size_t m = n - 1;
                    double dd = ((rthp[n] - rth_old[n]) - (rthp[m] - rth_old[m])) * (1.0 / G->dx);
                    double thf = (thL[n] + thL[m]) / 2.0;
                    rup[n] -= kap * dd / thf;
                }
                if (G->ty != FLAT) {
                    size_t m = n - STRY(G);
                    double dd = ((rthp[n] - rth_old[n]) - (rthp[m] - rth_old[m])) * (1.0 / G->dy);
                    double thf = (thL[n] + thL[m]) / 2.0;
                    rvp[n] -= kap * dd / thf;
                }
            }
}


/* apply_divergence_damping!(::DirectDivergenceDamping) (acoustic_substepping.jl:1158-1188): delta = V^-1 (dx(thetaF^x) + dy(thetaF^y))
 * with thetaF^x = Ax Ix(theta_L) (rho u)' (:962-963), halo fill of delta, then
 * (rho u)' += alpha dx^2 d_x(delta) / Ix(theta_L), (rho v)' += alpha dy^2 d_y(delta) / Iy(theta_L).  delta: scratch (density predictor). */
void og_direct_divergence_damping(const og_grid *G, double *rup, double *rvp, double *delta, const double *thL, double alpha)
{
    const ptrdiff_t sy = STRY(G);
#pragma omp parallel for collapse(2) schedule(static)
    for (int k = 0; k < G->Nz; ++k)
        for (int j = 0; j < G->Ny; ++j)
            for (int i = 0; i < G->Nx; ++i) {
                size_t n = IDX(G, i, j, k);
                double dzc = G->dzc[k + G->Hz];
                double Ax = G->dy * dzc, Ay = G->dx * dzc, Vinv = 1.0 / (G->dx * G->dy * dzc);
                double fx = 0.0, fy = 0.0;
                if (G->tx != FLAT)
                    fx = Ax * ((thL[n + 1] + thL[n]) / 2.0) * rup[n + 1] - Ax * ((thL[n] + thL[n - 1]) / 2.0) * rup[n];
                if (G->ty != FLAT)
                    fy = Ay * ((thL[n + sy] + thL[n]) / 2.0) * rvp[n + sy] - Ay * ((thL[n] + thL[n - sy]) / 2.0) * rvp[n];
                delta[n] = (fx + fy) * Vinv;
            }
    og_fill_halo_periodic_xy(G, delta, G->Nz + 2 * G->Hz);
    og_fill_halo_x_noflux(G, delta, G->Nz + 2 * G->Hz);      /* Bounded directions: the default (zero-gradient) condition of a centre field */
    og_fill_halo_y_noflux(G, delta, G->Nz + 2 * G->Hz);
#pragma omp parallel for collapse(2) schedule(static)
    for (int k = 0; k < G->Nz; ++k)
        for (int j = 0; j < G->Ny; ++j)
            for (int i = 0; i < G->Nx; ++i) {
                size_t n = IDX(G, i, j, k);
                if (G->tx != FLAT)
                    rup[n] += alpha * (G->dx * G->dx) * ((delta[n] - delta[n - 1]) * (1.0 / G->dx)) / ((thL[n] + thL[n - 1]) / 2.0);
                if (G->ty != FLAT)
                    rvp[n] += alpha * (G->dy * G->dy) * ((delta[n] - delta[n - sy]) * (1.0 / G->dy)) / ((thL[n] + thL[n - sy]) / 2.0);
            }
}

/* _relax_open_boundary_x! (dir 0) / _relax_open_boundary_y! (dir 1) (acoustic_substepping.jl:1323-1337): cb = outermost interior cell
 * (0-based: 0 or N - 1), ch = the adjacent halo cell (-1 or N) */
void og_relax_open_boundary(const og_grid *G, double *rp, double *rthp, const double *rho_d, const double *rth,
                            int dir, int cb, int ch, double alpha)
{
    int n1 = dir ? G->Nx : G->Ny;
    for (int k = 0; k < G->Nz; ++k)
        for (int t = 0; t < n1; ++t) {
            size_t nb = dir ? IDX(G, t, cb, k) : IDX(G, cb, t, k);
            size_t nh = dir ? IDX(G, t, ch, k) : IDX(G, ch, t, k);
            rp[nb] += alpha * ((rho_d[nh] - rho_d[nb]) / 2.0 - rp[nb]);
            rthp[nb] += alpha * ((rth[nh] - rth[nb]) / 2.0 - rthp[nb]);
        }
}

/* _zero_x_wall_face! (dir 0) / _zero_y_wall_face! (dir 1) (acoustic_substepping.jl:1367-1375): face = 0 or N (0-based; the reference's 1, N + 1) */
void og_zero_wall_face(const og_grid *G, double *f, int dir, int face)
{
    int n1 = dir ? G->Nx : G->Ny;
    for (int k = 0; k < G->Nz; ++k)
        for (int t = 0; t < n1; ++t)
            f[dir ? IDX(G, t, face, k) : IDX(G, face, t, k)] = 0.0;
}

/* _finalize_time_averaged_velocity! (acoustic_substepping.jl:1225-1250) */
void og_finalize_time_averaged_velocity(const og_grid *G, double *au, double *av, double *aw,
                                        const double *ru, const double *rv, const double *rw,
                                        const double *rho, double inv_N)
{
#pragma omp parallel for collapse(2) schedule(static)
    for (int k = 0; k < G->Nz; ++k)
        for (int j = 0; j < G->Ny; ++j)
            for (int i = 0; i < G->Nx; ++i) {
                size_t n = IDX(G, i, j, k);
                double rut = ru[n] + au[n] * inv_N;
                double rvt = rv[n] + av[n] * inv_N;
                double rwt = rw[n] + aw[n] * inv_N;
                double rx = (G->tx == FLAT) ? rho[n] : (rho[n] + rho[n - 1]) / 2.0;
                double ry = (G->ty == FLAT) ? rho[n] : (rho[n] + rho[n - STRY(G)]) / 2.0;
                double rz = (rho[n] + rho[n - STRZ(G)]) / 2.0;
                rx = (rx == 0.0) ? 1.0 : rx;
                ry = (ry == 0.0) ? 1.0 : ry;
                rz = (rz == 0.0) ? 1.0 : rz;
                au[n] = rut / rx;
                av[n] = rvt / ry;
                aw[n] = rwt / rz * (k > 0 ? 1.0 : 0.0);
            }
}

/* _recover_full_state! (acoustic_substepping.jl:1274-1292) */
void og_recover_full_state(const og_grid *G, double *rho_d, double *rtheta, double *ru, double *rv,
                           double *rw, const double *rp, const double *rthp, const double *rup,
                           const double *rvp, const double *rwp)
{
#pragma omp parallel for collapse(2) schedule(static)
    for (int k = 0; k < G->Nz; ++k)
        for (int j = 0; j < G->Ny; ++j)
            for (int i = 0; i < G->Nx; ++i) {
                size_t n = IDX(G, i, j, k);
                rho_d[n] = rho_d[n] + rp[n];
                rtheta[n] = rtheta[n] + rthp[n];
                ru[n] = ru[n] + rup[n];
                rv[n] = rv[n] + rvp[n];
                rw[n] = rw[n] + rwp[n];
            }
}

/* _rk3_substep! for non-acoustic scalars (acoustic_runge_kutta_3.jl:189-192) */
void og_ws_rk3_scalar(const og_grid *G, double *u, const double *u0, const double *Gn, double dt_stage)
{
#pragma omp parallel for collapse(2) schedule(static)
    for (int k = 0; k < G->Nz; ++k)
        for (int j = 0; j < G->Ny; ++j)
            for (int i = 0; i < G->Nx; ++i) {
                size_t n = IDX(G, i, j, k);
                u[n] = u0[n] + dt_stage * Gn[n];
            }
}
