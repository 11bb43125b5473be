// bz_fused.hip — the streaming half of the device-resident step, fused for HBM traffic:
//
//  k_rk3_march        ssp_rk3_substep! for the five prognostic fields; in the first stage (alpha = 1,
//                     where u = 0*u0 + (u + dt G)) it also performs store_initial_state! (U0 = u),
//                     so the separate copy pass disappears.
//                       /root/reference/src/TimeSteppers/ssp_runge_kutta_3.jl:114-186
//  k_project_diagnose make_pressure_correction! + compute_velocities! +
//                     compute_auxiliary_thermodynamic_variables! + every fill_halo_regions! that
//                     update_state! performs, in one pass: each thread also writes the periodic halo
//                     images and the z-halo copies of the values it produces, and scatters phi from the
//                     solver's contiguous buffer into the halo-inclusive pressure_anomaly field.
//                       /root/reference/src/AnelasticEquations/anelastic_time_stepping.jl:45-78
//                       /root/reference/src/AtmosphereModels/update_atmosphere_model_state.jl:41-68,122-155,225-292
//
// Launch shapes follow tools/membench.hip measurements on MI355X (5-field RK update, 512^3 + halos):
// 64x4 tiles per level 4.5 TB/s, 256x1 row blocks 5.3 TB/s, plane-contiguous 1-D indexing 5.9 TB/s.
// Pointwise kernels therefore index the contiguous run of interior rows of each z level (x halos
// included: they are overwritten by the next halo-image store), stencil kernels use 256x1 row blocks.
#include "bz_internal.h"

#define FX 64
#define FY 4

struct RKFieldsW {
    double *u[5];
    double *u0[5];        // written when FIRST
    const double *G[5];
};

template <bool FIRST>
__global__ __launch_bounds__(256) void k_rk3_rows(DevGrid g, RKFieldsW F, double dt, double alpha)
{
    // one z level per blockIdx.y; t runs over the contiguous rows j = 0..Ny-1 of that level
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= (long long)g.Ny * g.Sx) return;
    const int k = blockIdx.y;
    const long long n = g.Sxy * (k + g.Hz) + (long long)g.Hy * g.Sx + t;
    const double oma = 1.0 - alpha;
#pragma unroll
    for (int f = 0; f < 5; ++f) {
        if (f == 2 && k == 0) continue;          // rho_w wall face: never updated
        double u = F.u[f][n];
        if (FIRST) {
            F.u0[f][n] = u;
            F.u[f][n] = oma * u + alpha * (u + dt * F.G[f][n]);
        } else {
            F.u[f][n] = oma * F.u0[f][n] + alpha * (u + dt * F.G[f][n]);
        }
    }
}

__global__ __launch_bounds__(256) void k_poisson_source_rows(DevGrid g, double *__restrict__ rhs,
                                                             const double *__restrict__ ru,
                                                             const double *__restrict__ rv,
                                                             const double *__restrict__ rw, double dt, int k0)
{
    const int i = blockIdx.x * 256 + threadIdx.x, j = blockIdx.y, k = blockIdx.z + k0;
    if (i >= g.Nx) return;
    // periodic neighbours by wrap indexing: momentum halos need not be current here
    const long long ip = (i + 1 < g.Nx) ? 1 : 1 - g.Nx;
    // y-slab: row Ny is the neighbour rank's first row, delivered into the halo by the caller's exchange
    const long long jp = (j + 1 < g.Ny || !g.wrap_y) ? (long long)g.Sx : (long long)g.Sx * (1 - g.Ny);
    const long long n = g.idx(i, j, k);
    const double Ax = g.Ax[k], Ay = g.Ay[k], Az = g.Az;
    double a = Ax * ru[n + ip] - Ax * ru[n];
    double b = Ay * rv[n + jp] - Ay * rv[n];
    double c = Az * rw[n + g.Sxy] - Az * rw[n];
    double div = g.Vinv_c[k] * (a + b + c);
    rhs[(long long)i + (long long)g.Nx * ((long long)j + (long long)g.Ny * k)] = g.dzc[k] * div / dt;
}

// store v at n and at its periodic images (ox / oy = offset of the x / y image, 0 if none)
__device__ __forceinline__ void st_img(double *__restrict__ f, long long n, double v, long long ox, long long oy)
{
    f[n] = v;
    if (ox) f[n + ox] = v;
    if (oy) {
        f[n + oy] = v;
        if (ox) f[n + ox + oy] = v;
    }
}
// images only (the interior value is already in place)
__device__ __forceinline__ void st_img_only(double *__restrict__ f, long long n, double v, long long ox, long long oy)
{
    if (ox) f[n + ox] = v;
    if (oy) {
        f[n + oy] = v;
        if (ox) f[n + ox + oy] = v;
    }
}

// y image of row j: the periodic image (wrap_y), or — walls in y — the first halo row next to a wall row (no-flux copy of a centre-in-y
// field: the convention of the halo fill, bz_halo.hip); 0: none (interior rows; y-slabs, whose halos the neighbour ranks fill)
__device__ __forceinline__ long long bz_y_image(const DevGrid &g, int j)
{
    if (g.bounded_y) return (j == 0) ? -(long long)g.Sx : (j == g.Ny - 1) ? (long long)g.Sx : 0;
    if (!g.wrap_y) return 0;
    return (j < g.Hy) ? (long long)g.Ny * g.Sx : (j >= g.Ny - g.Hy) ? -(long long)g.Ny * g.Sx : 0;
}
// store of a y-face field (rho v, v): x image only; next to the north wall also the wall face j = Ny (zero) in the first halo row
__device__ __forceinline__ void st_yface(const DevGrid &g, double *__restrict__ f, long long n, double v, long long ox, int j)
{
    if (!g.bounded_y) return;
    f[n] = v;
    if (ox) f[n + ox] = v;
    if (j == g.Ny - 1) {
        f[n + g.Sx] = 0.0;
        if (ox) f[n + g.Sx + ox] = 0.0;
    }
}

// Block order of the two projection kernels (1-D launch of gx * Ny * nk workgroups).  They read phi at (i, j, k), (i-1, j, k),
// (i, j-1, k) and (i, j, k-1); launched as a (gx, Ny, nk) grid, the row below belonged to a workgroup on ANOTHER XCD (consecutive
// workgroup ids go round-robin to the 8 XCDs, each with its own L2), and the level below had been read a whole plane of seven
// arrays earlier: phi crossed the fabric three times (PMC: 9.65 GB per launch of k_project_lean for 7.52 GB of algorithmic
// traffic, r02_pmc_traffic.json).  Here XCD c owns the rows j in [c Ny/8, (c+1) Ny/8) of every level and walks them x fastest,
// then y, then z: the row below was read by the previous workgroups of the same XCD (except at its first row), the level below one
// eighth of a plane earlier (1.8 MB of traffic at 512^2 Float64: inside the 4 MB L2), and all XCDs stream through the same level.
#ifndef BZ_STREAM_BANDS_F32
#define BZ_STREAM_BANDS_F32 1
#endif
__device__ __forceinline__ void bz_stream_block(int gx, int Ny, int nk, int &bx, int &j, int &k)
{
    const unsigned w = blockIdx.x;
    unsigned r;
    // (Float32: round 3 measured plain launch order faster with one cell per thread, 0.87 against 0.95 ms for k_project_lean at 512^3; with two
    // cells per thread — one workgroup per 512-cell row — the band order wins, 0.910 -> 0.872 ms: BZ_STREAM_BANDS_F32)
    if ((sizeof(double) == 8 || BZ_STREAM_BANDS_F32) && (Ny & 7) == 0) {
        const unsigned c = w & 7u, rows = (unsigned)Ny >> 3;
        r = w >> 3;
        bx = (int)(r % (unsigned)gx); r /= (unsigned)gx;
        j = (int)(c * rows + r % rows);
        k = (int)(r / rows);
    } else {
        r = w;
        bx = (int)(r % (unsigned)gx); r /= (unsigned)gx;
        j = (int)(r % (unsigned)Ny);
        k = (int)(r / (unsigned)Ny);
    }
    bx = __builtin_amdgcn_readfirstlane(bx);
    j = __builtin_amdgcn_readfirstlane(j);
    k = __builtin_amdgcn_readfirstlane(k);
}

struct PDFields {
    double *ru, *rv, *rw, *rtheta, *rq;     // out (rtheta, rq: halos only, unless rtheta_in / rq_in differ: then the whole field)
    const double *rtheta_in, *rq_in;        // rho theta, rho q to diagnose from (the lean seam leaves them in the other ping-pong buffer)
    const double *ru_in, *rv_in, *rw_in;    // predictor momentum (= ru, rv, rw unless the RK update was fused into the tendencies)
    double *u, *v, *w, *theta, *q, *T;      // out
    double *phi;                            // out (halo-inclusive)
    const double *phi_c;                    // in: contiguous Nx*Ny*Nz, zero-mean solution
    const double *phi_below;                // y-slab only: phi of row j = -1 (neighbour rank), layout [k][i]
    int store_phi;                          // 0: skip the scatter into pressure_anomaly (stages whose phi nobody reads)
    int k0;                                 // first level of this launch (the chunked Poisson pipeline launches level ranges)
    // per-wave sums of the u, v, theta, q this launch stores, for the horizontal averages of SubsidenceForcing (bz_forcing.hip:
    // k_level_reduce): lsum[(f Nz + k) P + (j gx + bx) 4 + wave]; nullptr: not wanted.  Rows of a multiple of 64 cells only.
    double *lsum;
    long long lsum_P;
};

template <int SA>       // 0: no microphysics, 1: warm-phase saturation adjustment, 2: Kessler condensate species
__global__ __launch_bounds__(256) void k_project_diagnose(DevGrid g, PDFields F, double dt, int gx, int nk)
{
    int bx, j, k;
    bz_stream_block(gx, g.Ny, nk, bx, j, k);
    k += F.k0;
    const int i = bx * 256 + threadIdx.x;
    if (i >= g.Nx) return;
    const long long sz = g.Sxy;
    // periodic images of this column in the halo (requires Nx >= 2Hx, Ny >= 2Hy: at most one per direction)
    const long long ox = (i < g.Hx) ? g.Nx : (i >= g.Nx - g.Hx) ? -(long long)g.Nx : 0;
    const long long oy = bz_y_image(g, j);
    const bool wall = g.bounded_y && j == 0;      // walls in y: the wall face of rho v / v keeps its zero
    // contiguous-buffer neighbours (periodic wrap)
    const long long cplane = (long long)g.Nx * g.Ny;
    const long long m = (long long)i + (long long)g.Nx * j + cplane * k;
    const long long c_im = (i > 0) ? -1 : g.Nx - 1;
    const long long c_jm = (j > 0) ? -(long long)g.Nx : (long long)g.Nx * (g.Ny - 1);
    const long long n = g.idx(i, j, k);
    const bool bot = (k == 0), top = (k == g.Nz - 1);
    const double rc = g.rho[k], rf = g.rho_f[k];
    const double p = F.phi_c[m];
    const double p_im = F.phi_c[m + c_im];
    const double p_jm = wall ? p : (j == 0 && !g.wrap_y) ? F.phi_below[(long long)i + (long long)g.Nx * k] : F.phi_c[m + c_jm];

    // _pressure_correct_momentum!
    double ru = F.ru_in[n], rv = F.rv_in[n];
    ru -= rc * dt * ((p - p_im) * g.rdx);
    rv -= rc * dt * ((p - p_jm) * g.rdy);
    if (wall) rv = 0.0;
    // _compute_velocities!
    double u = ru / rc, v = rv / rc;
    // thermodynamic diagnosis
    const double rth = F.rtheta_in[n], rq = F.rq_in[n];
    const bool copy_scalars = (F.rtheta_in != F.rtheta);
    const double th = rth / rc, q = rq / rc;
    const double qd = 1.0 - q;
    const double Rm = qd * g.Rd + q * g.Rv;
    const double cpm = qd * g.cpd + q * g.cpv;
    double T, qvv = 0.0, qll = 0.0;
    double qcl = 0.0, qr = 0.0;
    if (SA == 2) {
        T = bz_kessler_T(g, th, q, g.qcl_field[n] + g.qr_field[n], g.p_r[k]);   // lagged condensate, as the reference (see k_thermo)
        qcl = g.rqcl_field[n] / rc;
        qr = g.rqr_field[n] / rc;
        qvv = q;
    } else if (SA == 1) T = bz_sa_diagnose(g, th, q, g.p_r[k], qvv, qll);
    else T = (g.formulation == 1) ? (th - g.g * g.zc[k]) / cpm : bz_exner_factor(g, k, q, cpm) * th;

    if (F.lsum) {      // the stage's horizontal sums ride on the values in registers (a separate pass reads the four arrays again)
        double a0 = u, a1 = v, a2 = th, a3 = q;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            a0 += __shfl_down(a0, off); a1 += __shfl_down(a1, off); a2 += __shfl_down(a2, off); a3 += __shfl_down(a3, off);
        }
        if ((threadIdx.x & 63) == 0) {
            const long long slot = ((long long)j * gx + bx) * 4 + (threadIdx.x >> 6), fs = (long long)g.Nz * F.lsum_P;
            double *o = F.lsum + (long long)k * F.lsum_P + slot;
            o[0] = a0; o[fs] = a1; o[2 * fs] = a2; o[3 * fs] = a3;
        }
    }

    if (F.store_phi) st_img(F.phi, n, p, ox, oy);
    st_img(F.ru, n, ru, ox, oy);
    if (g.bounded_y) { st_yface(g, F.rv, n, rv, ox, j); st_yface(g, F.v, n, v, ox, j); }
    else { st_img(F.rv, n, rv, ox, oy); st_img(F.v, n, v, ox, oy); }
    st_img(F.u, n, u, ox, oy);
    st_img(F.theta, n, th, ox, oy);
    st_img(F.q, n, q, ox, oy);
    st_img(F.T, n, T, ox, oy);
    if (SA == 1) {
        st_img(g.qv_field, n, qvv, ox, oy);
        st_img(g.ql_field, n, qll, ox, oy);
    }
    if (SA == 2) {
        st_img(g.qv_field, n, qvv, ox, oy);
        st_img(g.qcl_field, n, qcl, ox, oy);
        st_img(g.qr_field, n, qr, ox, oy);
        st_img_only(g.rqcl_field, n, g.rqcl_field[n], ox, oy);
        st_img_only(g.rqr_field, n, g.rqr_field[n], ox, oy);
    }
    if (copy_scalars) {
        st_img(F.rtheta, n, rth, ox, oy);
        st_img(F.rq, n, rq, ox, oy);
    } else {
        st_img_only(F.rtheta, n, rth, ox, oy);
        st_img_only(F.rq, n, rq, ox, oy);
    }
    if (!bot) {      // wall face k = 0 keeps rho_w = w = 0
        const double p_km = F.phi_c[m - cplane];
        double rw = F.rw_in[n];
        rw -= rf * dt * ((p - p_km) * g.rdzf[k]);
        st_img(F.rw, n, rw, ox, oy);
        st_img(F.w, n, rw / rf, ox, oy);
    }
    if (bot || top) {    // first z-halo cell of no-flux centre fields; u, v: level Nz only
        const long long h = bot ? -sz : sz;
        if (F.store_phi) st_img(F.phi, n + h, p, ox, oy);
        st_img(F.ru, n + h, ru, ox, oy);
        if (g.bounded_y) st_yface(g, F.rv, n + h, rv, ox, j);
        else st_img(F.rv, n + h, rv, ox, oy);
        st_img(F.theta, n + h, th, ox, oy);
        st_img(F.q, n + h, q, ox, oy);
        st_img(F.T, n + h, T, ox, oy);
        if (SA == 1) {
            st_img(g.qv_field, n + h, qvv, ox, oy);
            st_img(g.ql_field, n + h, qll, ox, oy);
        }
        if (SA == 2) {
            st_img(g.qv_field, n + h, qvv, ox, oy);
            st_img(g.qcl_field, n + h, qcl, ox, oy);
            st_img(g.qr_field, n + h, qr, ox, oy);
            st_img(g.rqcl_field, n + h, g.rqcl_field[n], ox, oy);
            st_img(g.rqr_field, n + h, g.rqr_field[n], ox, oy);
        }
        st_img(F.rtheta, n + h, rth, ox, oy);
        st_img(F.rq, n + h, rq, ox, oy);
        if (top) {
            st_img(F.u, n + sz, u, ox, oy);
            if (g.bounded_y) st_yface(g, F.v, n + sz, v, ox, j);
            else st_img(F.v, n + sz, v, ox, oy);
        }
    }
}


// make_pressure_correction! alone, for the stages of the lean whole-step seam whose diagnostics nobody reads (bz_step.hip):
// projected momentum + its periodic images and z-halo copies.  7 words per cell instead of 18.
struct PLFields {
    double *ru, *rv, *rw;
    const double *ru_in, *rv_in, *rw_in;
    const double *phi_c;
    const double *phi_below;
    double *sa, *sb;             // rho theta, rho q just advanced by the lean scalar kernel (interior): their periodic images are stored here
    int k0;                      // first level of this launch
};
// PV consecutive x cells per thread: 1 in Float64; 2 in the Float32 build, where a 4-byte request per lane moves 256 bytes per wave
// instruction — the Float32 kernel streamed at 3.7 TB/s against 5.5 TB/s for the same kernel in Float64 (round 4: 8 bytes per lane)
#ifndef BZ_PV
#define BZ_PV (sizeof(double) == 8 ? 1 : 2)
#endif
typedef double bz_pv2 __attribute__((ext_vector_type(2), aligned(sizeof(double))));      // two consecutive reals, element-aligned
template <int PV> __device__ __forceinline__ void pv_load(const double *__restrict__ p, double (&v)[PV])
{
    if constexpr (PV == 2) { const bz_pv2 t = *(const bz_pv2 *)p; v[0] = t.x; v[1] = t.y; }
    else v[0] = p[0];
}
template <int PV> __device__ __forceinline__ void pv_store(double *__restrict__ p, const double (&v)[PV])
{
    if constexpr (PV == 2) { bz_pv2 t; t.x = v[0]; t.y = v[1]; *(bz_pv2 *)p = t; }
    else p[0] = v[0];
}

template <int PV>
__global__ __launch_bounds__(256) void k_project_lean(DevGrid g, PLFields F, double dt, int gx, int nk)
{
    int bx, j, k;
    bz_stream_block(gx, g.Ny, nk, bx, j, k);
    k += F.k0;
    const int i = (bx * 256 + threadIdx.x) * PV;      // first of the thread's PV cells (Nx is a multiple of PV: the launcher checks)
    if (i >= g.Nx) return;
    const long long sz = g.Sxy;
    const long long oy = bz_y_image(g, j);
    const bool wall = g.bounded_y && j == 0;      // walls in y: the wall face of rho v keeps its zero
    const long long cplane = (long long)g.Nx * g.Ny;
    const long long m = (long long)i + (long long)g.Nx * j + cplane * k;
    const long long c_im = (i > 0) ? -1 : g.Nx - 1;
    const long long c_jm = (j > 0) ? -(long long)g.Nx : (long long)g.Nx * (g.Ny - 1);
    const long long n = g.idx(i, j, k);
    const bool bot = (k == 0), top = (k == g.Nz - 1);
    const double rc = g.rho[k], rf = g.rho_f[k];
    double p[PV], p_jm[PV], p_km[PV], ru[PV], rv[PV], rw[PV];
    pv_load<PV>(F.phi_c + m, p);
    const double p_left = F.phi_c[m + c_im];
    if (wall) { for (int c = 0; c < PV; ++c) p_jm[c] = p[c]; }
    else if (j == 0 && !g.wrap_y) pv_load<PV>(F.phi_below + (long long)i + (long long)g.Nx * k, p_jm);
    else pv_load<PV>(F.phi_c + m + c_jm, p_jm);
    pv_load<PV>(F.ru_in + n, ru);
    pv_load<PV>(F.rv_in + n, rv);
    if (!bot) { pv_load<PV>(F.phi_c + m - cplane, p_km); pv_load<PV>(F.rw_in + n, rw); }
#pragma unroll
    for (int c = 0; c < PV; ++c) {
        const double p_im = (c == 0) ? p_left : p[c > 0 ? c - 1 : 0];
        ru[c] -= rc * dt * ((p[c] - p_im) * g.rdx);
        rv[c] -= rc * dt * ((p[c] - p_jm[c]) * g.rdy);
        if (wall) rv[c] = 0.0;
        if (!bot) rw[c] -= rf * dt * ((p[c] - p_km[c]) * g.rdzf[k]);
    }
    // interior values as one request per field, then the halo images of the cells that have any (x images: the first / last Hx cells of a row)
    pv_store<PV>(F.ru + n, ru);
    if (!g.bounded_y) pv_store<PV>(F.rv + n, rv);
    if (!bot) pv_store<PV>(F.rw + n, rw);
    if (bot || top) {
        const long long h = bot ? -sz : sz;
        pv_store<PV>(F.ru + n + h, ru);
        if (!g.bounded_y) pv_store<PV>(F.rv + n + h, rv);
    }
#pragma unroll
    for (int c = 0; c < PV; ++c) {
        const int ic = i + c;
        const long long nc = n + c;
        const long long ox = (ic < g.Hx) ? g.Nx : (ic >= g.Nx - g.Hx) ? -(long long)g.Nx : 0;
        if (g.bounded_y) st_yface(g, F.rv, nc, rv[c], ox, j);
        if (ox | oy) {
            st_img_only(F.ru, nc, ru[c], ox, oy);
            if (!g.bounded_y) st_img_only(F.rv, nc, rv[c], ox, oy);
            if (!bot) st_img_only(F.rw, nc, rw[c], ox, oy);
            // edge cells only: the halo images of the scalars
            st_img_only(F.sa, nc, F.sa[nc], ox, oy);
            st_img_only(F.sb, nc, F.sb[nc], ox, oy);
        }
        if (bot || top) {
            const long long h = bot ? -sz : sz;
            if (ox | oy) { st_img_only(F.ru, nc + h, ru[c], ox, oy); if (!g.bounded_y) st_img_only(F.rv, nc + h, rv[c], ox, oy); }
            if (g.bounded_y) st_yface(g, F.rv, nc + h, rv[c], ox, j);
        }
    }
}

#include "bz_xfft_kernels.h"

static size_t xf_lds_bytes(int n2) { return ((size_t)XF_RB * XF_ROW_STRIDE(n2) + xf_w_entries(n2) + XF_WST_SLOTS(n2)) * sizeof(double2); }
// levels one block of the x-transform kernels walks: 16 on large grids (the twiddle table is staged once per block), fewer when
// that would leave less than ~1024 blocks for the 256 CUs (64^3: 32 blocks of 16 levels took 52 us, 512 blocks of one level 1/3 of that)
static int xf_chunk(int forced, int dflt, const DevGrid &g)
{
    int kc = dflt;
    const long long rows = (long long)(g.Ny / XF_RB) * g.Nz;
    while (kc > 1 && rows / kc < 1024) kc >>= 1;
    if (forced > 0) kc = forced;
    return kc < g.Nz ? kc : g.Nz;
}

// x transform of the rows of the source term into the transposed half spectrum `hat` (nullptr: ctx->d_hat); predictor != nullptr
// (or s != nullptr): the source term is evaluated on the fly from that momentum, else the rows come from rhs (nullptr: ctx->d_rhs).
// blocks > 1 (y-slab ranks): `hat` is `blocks` messages of ctx->nkx wavenumbers each (XfLayout).
int bzi_xf_forward(bz_ctx *ctx, const bz_state *s, double dt, const bz_prognostic *predictor, const double *rhs, double *hat, int blocks, int klo,
                   int khi)
{
    const DevGrid &g = ctx->dg;
    if (khi <= 0) { klo = 0; khi = g.Nz; }      // whole column
    const int n2 = g.Nx / 2, kc = xf_chunk(ctx->tune.xf_kchunk_f, 16, g);
    const dim3 grid(g.Ny / XF_RB, (khi - klo + kc - 1) / kc), block(XF_RB * (n2 / 4));
    const size_t lds = xf_lds_bytes(n2);
    XfLayout L;
    L.nkx = blocks > 1 ? ctx->nkx : n2 + 1;
    L.nxp = blocks > 1 ? ctx->nkx * blocks : n2 + 1;
    L.blk = (long long)g.Nz * L.nkx * g.Ny;
    L.klo = klo; L.khi = khi;
    L.kxs = (ctx->kxmajor && blocks == 1 && !hat) ? (long long)(g.Nz + ctx->kx_pad) * g.Ny : 0;
    double2 *out = (double2 *)(hat ? hat : (double *)ctx->d_hat);
    const double *pu = predictor ? predictor->rho_u : s ? s->rho_u : nullptr, *pv = predictor ? predictor->rho_v : s ? s->rho_v : nullptr,
                 *pw = predictor ? predictor->rho_w : s ? s->rho_w : nullptr;
    const double *rows = (s || predictor) ? nullptr : (rhs ? rhs : (const double *)ctx->d_rhs);
    const double2 *W = (const double2 *)ctx->d_wtab;
    if (n2 / 4 > 64 || n2 % 3 == 0) {      // Nx = 1024: teams of two wavefronts, 78 KB of LDS per workgroup; Nx = 3 * 2^m: teams of 12 ... 96 threads straddle wavefronts — workgroup barriers between the stages
        static bool once[2] = {false, false};
        if (!once[0]) { BZ_HIP(hipFuncSetAttribute((const void *)k_x_forward<1, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); once[0] = true; }
        if (!once[1]) { BZ_HIP(hipFuncSetAttribute((const void *)k_x_forward<0, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); once[1] = true; }
        if (rows) hipLaunchKernelGGL((k_x_forward<0, 2>), grid, block, lds, ctx->stream, g, rows, pu, pv, pw, dt, out, L, W, kc);
        else hipLaunchKernelGGL((k_x_forward<1, 2>), grid, block, lds, ctx->stream, g, rows, pu, pv, pw, dt, out, L, W, kc);
    } else if (rows) hipLaunchKernelGGL((k_x_forward<0, 1>), grid, block, lds, ctx->stream, g, rows, pu, pv, pw, dt, out, L, W, kc);
    else hipLaunchKernelGGL((k_x_forward<1, 1>), grid, block, lds, ctx->stream, g, rows, pu, pv, pw, dt, out, L, W, kc);
    BZ_LAUNCH_CHECK();
    return BZ_OK;
}

// inverse x transform of the transposed half spectrum `hat` (nullptr: ctx->d_hat; blocks as above): phi into `phi` (nullptr: ctx->d_rhs)
int bzi_xf_inverse(bz_ctx *ctx, const double *hat, double *phi, int blocks, int klo, int khi)
{
    const DevGrid &g = ctx->dg;
    if (khi <= 0) { klo = 0; khi = g.Nz; }
    const int n2 = g.Nx / 2, kc = xf_chunk(ctx->tune.xf_kchunk_i, 16, g);
    XfLayout L;
    L.nkx = blocks > 1 ? ctx->nkx : n2 + 1;
    L.nxp = blocks > 1 ? ctx->nkx * blocks : n2 + 1;
    L.blk = (long long)g.Nz * L.nkx * g.Ny;
    L.klo = klo; L.khi = khi;
    L.kxs = (ctx->kxmajor && blocks == 1 && !hat) ? (long long)(g.Nz + ctx->kx_pad) * g.Ny : 0;
    const double2 *in = (const double2 *)(hat ? hat : (const double *)ctx->d_hat);
    double *out = phi ? phi : ctx->d_rhs;
    const dim3 grid(g.Ny / XF_RB, (khi - klo + kc - 1) / kc), block(XF_RB * (n2 / 4));
    const size_t lds = xf_lds_bytes(n2);
    if (n2 / 4 > 64 || n2 % 3 == 0) {
        static bool once = false;
        if (!once) { BZ_HIP(hipFuncSetAttribute((const void *)k_x_inverse<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); once = true; }
        hipLaunchKernelGGL(k_x_inverse<2>, grid, block, lds, ctx->stream, g, in, L, out, (const double2 *)ctx->d_wtab, kc);
    } else hipLaunchKernelGGL(k_x_inverse<1>, grid, block, lds, ctx->stream, g, in, L, out, (const double2 *)ctx->d_wtab, kc);
    BZ_LAUNCH_CHECK();
    return BZ_OK;
}

// NOTE (y-slab mode): the periodic images stored by k_project_diagnose cover x and z only; the caller exchanges
// the y halos of the fields the tendencies read (bz_* slab entry points in bz_slab.hip).
int bzi_rk3_fused(bz_ctx *ctx, const bz_state *s, const bz_prognostic *U0, const bz_prognostic *G, double dt,
                  double alpha, bool first)
{
    const DevGrid &g = ctx->dg;
    ProfileScope ps(ctx, first ? "ssp_rk3_substep+store_initial_state" : "ssp_rk3_substep");
    RKFieldsW F;
    F.u[0] = s->rho_u; F.u[1] = s->rho_v; F.u[2] = s->rho_w; F.u[3] = s->rho_theta; F.u[4] = s->rho_q;
    F.u0[0] = U0->rho_u; F.u0[1] = U0->rho_v; F.u0[2] = U0->rho_w; F.u0[3] = U0->rho_theta; F.u0[4] = U0->rho_q;
    F.G[0] = G->rho_u; F.G[1] = G->rho_v; F.G[2] = G->rho_w; F.G[3] = G->rho_theta; F.G[4] = G->rho_q;
    long long per_level = (long long)g.Ny * g.Sx;
    dim3 grid((unsigned)((per_level + 255) / 256), g.Nz), block(256);
    if (first)
        hipLaunchKernelGGL(k_rk3_rows<true>, grid, block, 0, ctx->stream, g, F, dt, alpha);
    else
        hipLaunchKernelGGL(k_rk3_rows<false>, grid, block, 0, ctx->stream, g, F, dt, alpha);
    BZ_LAUNCH_CHECK();
    return BZ_OK;
}

int bzi_poisson_source_fused(bz_ctx *ctx, const bz_state *s, double dt, double *rhs, const bz_prognostic *predictor)
{
    const DevGrid &g = ctx->dg;
    ProfileScope ps(ctx, "poisson_source_term");
    const int k0 = ctx->krn ? ctx->kr0 : 0, nk = ctx->krn ? ctx->krn : g.Nz;      // level range (chunked Poisson pipeline) or all
    dim3 grid((g.Nx + 255) / 256, g.Ny, nk), block(256);
    hipLaunchKernelGGL(k_poisson_source_rows, grid, block, 0, ctx->stream, g, rhs ? rhs : ctx->d_rhs,
                       predictor ? predictor->rho_u : s->rho_u, predictor ? predictor->rho_v : s->rho_v,
                       predictor ? predictor->rho_w : s->rho_w, dt, k0);
    BZ_LAUNCH_CHECK();
    return BZ_OK;
}

int bzi_project_lean(bz_ctx *ctx, const bz_state *s, double dt, const double *phi_c, const double *phi_below,
                     const bz_prognostic *predictor, double *sa, double *sb)
{
    const DevGrid &g = ctx->dg;
    ProfileScope ps(ctx, "project_momentum");
    PLFields F;
    F.ru = s->rho_u; F.rv = s->rho_v; F.rw = s->rho_w;
    F.ru_in = predictor->rho_u; F.rv_in = predictor->rho_v; F.rw_in = predictor->rho_w;
    F.phi_c = phi_c ? phi_c : ctx->d_rhs;
    F.phi_below = phi_below;
    F.sa = sa; F.sb = sb;
    F.k0 = ctx->krn ? ctx->kr0 : 0;
    const int nk = ctx->krn ? ctx->krn : g.Nz;
    if (BZ_PV == 2 && g.Nx % 2 == 0) {
        const int gx = (g.Nx / 2 + 255) / 256;
        dim3 grid((unsigned)((long long)gx * g.Ny * nk)), block(256);
        hipLaunchKernelGGL(k_project_lean<2>, grid, block, 0, ctx->stream, g, F, dt, gx, nk);
    } else {
        const int gx = (g.Nx + 255) / 256;
        dim3 grid((unsigned)((long long)gx * g.Ny * nk)), block(256);
        hipLaunchKernelGGL(k_project_lean<1>, grid, block, 0, ctx->stream, g, F, dt, gx, nk);
    }
    BZ_LAUNCH_CHECK();
    return BZ_OK;
}

int bzi_project_diagnose(bz_ctx *ctx, const bz_state *s, double dt, const double *phi_c, const double *phi_below,
                         const bz_prognostic *predictor, bool store_phi, const double *rtheta_in, const double *rq_in, bool level_sums)
{
    const DevGrid &g = ctx->dg;
    ProfileScope ps(ctx, "project_and_diagnose");
    PDFields F;
    F.ru = s->rho_u; F.rv = s->rho_v; F.rw = s->rho_w; F.rtheta = s->rho_theta; F.rq = s->rho_q;
    F.rtheta_in = rtheta_in ? rtheta_in : s->rho_theta; F.rq_in = rq_in ? rq_in : s->rho_q;
    F.ru_in = predictor ? predictor->rho_u : s->rho_u;
    F.rv_in = predictor ? predictor->rho_v : s->rho_v;
    F.rw_in = predictor ? predictor->rho_w : s->rho_w;
    F.u = s->u; F.v = s->v; F.w = s->w; F.theta = s->theta; F.q = s->q; F.T = s->T;
    F.phi = s->phi;
    F.phi_c = phi_c ? phi_c : ctx->d_rhs;
    F.phi_below = phi_below;
    F.store_phi = store_phi ? 1 : 0;
    F.k0 = ctx->krn ? ctx->kr0 : 0;
    const int gx = (g.Nx + 255) / 256, nk = ctx->krn ? ctx->krn : g.Nz;
    F.lsum = level_sums ? bzi_level_sum_rows(ctx, &F.lsum_P) : nullptr;
    if (!F.lsum) F.lsum_P = 0;
    dim3 grid((unsigned)((long long)gx * g.Ny * nk)), block(256);
    if (g.microphysics == 2)
        hipLaunchKernelGGL((k_project_diagnose<2>), grid, block, 0, ctx->stream, g, F, dt, gx, nk);
    else if (g.microphysics == 1)
        hipLaunchKernelGGL((k_project_diagnose<1>), grid, block, 0, ctx->stream, g, F, dt, gx, nk);
    else
        hipLaunchKernelGGL((k_project_diagnose<0>), grid, block, 0, ctx->stream, g, F, dt, gx, nk);
    BZ_LAUNCH_CHECK();
    return BZ_OK;
}
