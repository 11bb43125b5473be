// bz_slab.hip — entry points for the y-slab (one process per GPU) decomposition.  The collective steps —
// y-halo exchange and the transposes of the distributed Fourier transform — are done by the host
// (breeze.jl_amd/distributed.py, torch.distributed over RCCL/xGMI); these calls are the rank-local kernels
// in between.  They are the same kernels the single-GPU whole-step seam uses (bz_fused.hip, bz_poisson.hip).
#include "bz_internal.h"

__global__ void k_tridiag_solve(int NXH, int Ny, int Nz, const double *__restrict__ lower,
                                const double *__restrict__ ibeta, const double *__restrict__ tfac,
                                double2 *__restrict__ hat, double scale, int mean_column);

extern "C" int bz_slab_info(bz_ctx *ctx, int32_t *y_nranks, int32_t *y_rank, int32_t *nkx, int32_t *kx0, int32_t *ny_global)
{
    if (!ctx) return BZ_ERR_INVALID;
    if (y_nranks) *y_nranks = ctx->y_nranks;
    if (y_rank) *y_rank = ctx->y_rank;
    if (nkx) *nkx = ctx->nkx;
    if (kx0) *kx0 = ctx->kx0;
    if (ny_global) *ny_global = ctx->Ny_global;
    return BZ_OK;
}

extern "C" int bz_ssp_rk3_substep_fused(bz_ctx *ctx, const bz_state *s, const bz_prognostic *U0,
                                        const bz_prognostic *G, double dt, double alpha, int first)
{
    if (!ctx || !s || !U0 || !G) return BZ_ERR_INVALID;
    if (ctx->G_is_predictor) {
        int rc = bz_compute_tendencies(ctx, s, G);
        if (rc) return rc;
    }
    return bzi_rk3_fused(ctx, s, U0, G, dt, alpha, first != 0);
}

extern "C" int bz_poisson_source_term(bz_ctx *ctx, const bz_state *s, double dt, double *rhs)
{
    if (!ctx || !s || !rhs) return BZ_ERR_INVALID;
    return bzi_poisson_source_fused(ctx, s, dt, rhs);
}

extern "C" int bz_spectral_tridiagonal_solve(bz_ctx *ctx, double *hat, double scale)
{
    if (!ctx || !hat) return BZ_ERR_INVALID;
    ProfileScope ps(ctx, "poisson_tridiagonal");
    const int Ny = ctx->slab_mode ? ctx->Ny_global : ctx->dg.Ny;
    return bzi_tridiag_launch(ctx, hat, scale, Ny, ctx->kx0 == 0 ? 1 : 0);
}

// Transposing pack of the distributed transform: out[k][b][a] = in[k][a][c0 + b] for a < A, b < B (zero where c0 + b >= valid),
// complex Float64.  `in` is (Nz, A, ld), `out` (Nz, B, A).  32 x 32 tiles through LDS so that both the read (along the last
// axis of `in`) and the write (along the last axis of `out`) are coalesced; this replaces the strided elementwise copy of
// tensor.permute(0, 2, 1).contiguous(), measured at 2 TB/s, and the zero padding of the half spectrum.
__global__ __launch_bounds__(256) void k_pack_transpose(const double2 *__restrict__ in, double2 *__restrict__ out, int A, int ld,
                                                        int c0, int B, int valid)
{
    __shared__ double2 tile[32][33];
    const int k = blockIdx.z;
    const int a0 = blockIdx.y * 32, b0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;          // 32 x 8
    const double2 *src = in + (long long)k * A * ld;
    double2 *dst = out + (long long)k * B * A;
#pragma unroll
    for (int r = 0; r < 32; r += 8) {
        const int a = a0 + ty + r, b = b0 + tx;
        double2 v = make_double2(0.0, 0.0);
        if (a < A && b < B && c0 + b < valid) v = src[(long long)a * ld + c0 + b];
        tile[ty + r][tx] = v;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 32; r += 8) {
        const int b = b0 + ty + r, a = a0 + tx;
        if (a < A && b < B) dst[(long long)b * A + a] = tile[tx][ty + r];
    }
}

extern "C" int bz_pack_transpose(bz_ctx *ctx, const double *in, double *out, int32_t Nz, int32_t A, int32_t ld, int32_t c0,
                                 int32_t B, int32_t valid)
{
    if (!ctx || !in || !out || Nz < 1 || A < 1 || B < 1 || ld < 1 || c0 < 0) return BZ_ERR_INVALID;
    ProfileScope ps(ctx, "poisson_transpose_pack");
    hipLaunchKernelGGL(k_pack_transpose, dim3((B + 31) / 32, (A + 31) / 32, Nz), dim3(256), 0, ctx->stream,
                       (const double2 *)in, (double2 *)out, A, ld, c0, B, valid);
    BZ_LAUNCH_CHECK();
    return BZ_OK;
}

// The four local transforms of the distributed Poisson solve as 1-D batched rocFFT plans (no normalisation anywhere; 1/(Nx Ny)
// rides on bz_spectral_tridiagonal_solve's scale):
//   which 0  x forward   rhs (Nz, Ny, Nx) real            -> out (Nz, Ny, Nx/2+1) complex
//   which 1  y forward   in place on (Nz, nkx, Ny_global) complex (in == out)
//   which 2  y backward  in place
//   which 3  x backward  in (Nz, Ny, ld) complex, ld >= Nx/2+1 (the padded half spectrum) -> out (Nz, Ny, Nx) real; destroys `in`
static int slab_plans(bz_ctx *ctx, int ld)
{
    const DevGrid &g = ctx->dg;
    if (ctx->slab_plans_ok && ctx->slab_inv_ld == ld) return BZ_OK;
    if (ctx->slab_plans_ok) {
        hipfftDestroy(ctx->slab_plan_x_fwd); hipfftDestroy(ctx->slab_plan_x_inv); hipfftDestroy(ctx->slab_plan_y);
        ctx->slab_plans_ok = false;
    }
    const int nxh = g.Nx / 2 + 1;
    int nx[1] = {g.Nx}, ny[1] = {ctx->Ny_global};
    int real_embed[1] = {g.Nx}, half_embed[1] = {nxh}, pad_embed[1] = {ld}, y_embed[1] = {ctx->Ny_global};
    BZ_FFT(hipfftPlanMany(&ctx->slab_plan_x_fwd, 1, nx, real_embed, 1, g.Nx, half_embed, 1, nxh, HIPFFT_D2Z, g.Nz * g.Ny));
    BZ_FFT(hipfftPlanMany(&ctx->slab_plan_x_inv, 1, nx, pad_embed, 1, ld, real_embed, 1, g.Nx, HIPFFT_Z2D, g.Nz * g.Ny));
    BZ_FFT(hipfftPlanMany(&ctx->slab_plan_y, 1, ny, y_embed, 1, ctx->Ny_global, y_embed, 1, ctx->Ny_global, HIPFFT_Z2Z, g.Nz * ctx->NXH));
    BZ_FFT(hipfftSetStream(ctx->slab_plan_x_fwd, ctx->stream));
    BZ_FFT(hipfftSetStream(ctx->slab_plan_x_inv, ctx->stream));
    BZ_FFT(hipfftSetStream(ctx->slab_plan_y, ctx->stream));
    ctx->slab_plans_ok = true;
    ctx->slab_inv_ld = ld;
    return BZ_OK;
}

extern "C" int bz_slab_transform(bz_ctx *ctx, int32_t which, double *in, double *out, int32_t ld)
{
    if (!ctx || !in || !out || which < 0 || which > 3) return BZ_ERR_INVALID;
    if (!ctx->slab_mode) { ctx->last_error = "bz_slab_transform: y-slab contexts only"; return BZ_ERR_UNSUPPORTED; }
    const int nxh = ctx->dg.Nx / 2 + 1;
    if (which == 3 && ld < nxh) return BZ_ERR_INVALID;
    int rc = slab_plans(ctx, which == 3 ? ld : (ctx->slab_plans_ok ? ctx->slab_inv_ld : nxh));
    if (rc) return rc;
    ProfileScope ps(ctx, which == 0 ? "poisson_fft_x_forward" : (which == 3 ? "poisson_fft_x_inverse" : "poisson_fft_y"));
    switch (which) {
        case 0: BZ_FFT(hipfftExecD2Z(ctx->slab_plan_x_fwd, in, (hipfftDoubleComplex *)out)); break;
        case 1: BZ_FFT(hipfftExecZ2Z(ctx->slab_plan_y, (hipfftDoubleComplex *)in, (hipfftDoubleComplex *)out, HIPFFT_FORWARD)); break;
        case 2: BZ_FFT(hipfftExecZ2Z(ctx->slab_plan_y, (hipfftDoubleComplex *)in, (hipfftDoubleComplex *)out, HIPFFT_BACKWARD)); break;
        default: BZ_FFT(hipfftExecZ2D(ctx->slab_plan_x_inv, (hipfftDoubleComplex *)in, out)); break;
    }
    return BZ_OK;
}

// y-halo exchange helpers: gather (unpack == 0) / scatter (unpack != 0) `nrows` rows starting at parent row `row0` of up to
// BZ_MAX_ROW_FIELDS parent arrays (all z levels, full x width) into / out of one contiguous buffer — one launch per direction
// instead of one strided copy per field.  Buffer layout: field-major, then (level, row, x).
#define BZ_MAX_ROW_FIELDS 24
struct RowFields {
    double *f[BZ_MAX_ROW_FIELDS];
    int levels[BZ_MAX_ROW_FIELDS];
    long long offset[BZ_MAX_ROW_FIELDS];      // start of the field's block in the buffer (elements)
    int n;
};

__global__ __launch_bounds__(256) void k_row_pack(RowFields R, double *__restrict__ buf, int Sx, long long Sxy, int row0, int nrows,
                                                  int unpack)
{
    const int m = blockIdx.z;
    const int per_level = nrows * Sx;
    const long long total = (long long)R.levels[m] * per_level;
    double *f = R.f[m];
    double *b = buf + R.offset[m];
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const int lev = (int)(e / per_level), rem = (int)(e % per_level);
        const long long n = Sxy * lev + (long long)(row0 + rem / Sx) * Sx + rem % Sx;
        if (unpack) f[n] = b[e];
        else b[e] = f[n];
    }
}

// sx / sxy: row length and plane stride of the fields in DOUBLES — g.Sx / g.Sxy for fields of the grid's real; half of them for the
// Float32 working fields of a Float64 model (substep_floattype = Float32 on y-slabs: a row of Sx floats travels as Sx / 2 doubles)
int bzi_pack_rows_geom(bz_ctx *ctx, double *const *fields, const int32_t *levels, int32_t n, int32_t row0, int32_t nrows, double *buffer,
                       int32_t unpack, int sx, long long sxy)
{
    if (!ctx || !fields || !levels || !buffer || n < 1 || n > BZ_MAX_ROW_FIELDS || nrows < 1 || row0 < 0) return BZ_ERR_INVALID;
    RowFields R;
    R.n = n;
    long long off = 0, maxtot = 0;
    for (int m = 0; m < n; ++m) {
        if (!fields[m] || levels[m] < 1) return BZ_ERR_INVALID;
        R.f[m] = fields[m];
        R.levels[m] = levels[m];
        R.offset[m] = off;
        const long long tot = (long long)levels[m] * nrows * sx;
        off += tot;
        if (tot > maxtot) maxtot = tot;
    }
    ProfileScope ps(ctx, unpack ? "halo_rows_unpack" : "halo_rows_pack");
    const unsigned bx = (unsigned)((maxtot + 255) / 256 > 4096 ? 4096 : (maxtot + 255) / 256);
    hipLaunchKernelGGL(k_row_pack, dim3(bx, 1, n), dim3(256), 0, ctx->stream, R, buffer, sx, sxy, row0, nrows, unpack);
    BZ_LAUNCH_CHECK();
    return BZ_OK;
}

extern "C" int bz_pack_rows(bz_ctx *ctx, double *const *fields, const int32_t *levels, int32_t n, int32_t row0, int32_t nrows,
                            double *buffer, int32_t unpack)
{
    if (!ctx) return BZ_ERR_INVALID;
    return bzi_pack_rows_geom(ctx, fields, levels, n, row0, nrows, buffer, unpack, ctx->dg.Sx, ctx->dg.Sxy);
}

extern "C" int bz_project_and_diagnose(bz_ctx *ctx, const bz_state *s, const double *phi_c, const double *phi_below,
                                       double dt)
{
    if (!ctx || !s || !phi_c) return BZ_ERR_INVALID;
    if (!ctx->dg.wrap_y && !phi_below) return BZ_ERR_INVALID;
    return bzi_project_diagnose(ctx, s, dt, phi_c, phi_below);
}

// compute_tendencies! with the following ssp_rk3_substep! folded in (see bz_time_step_anelastic): predictor
// momentum -> G->rho_u/v/w, rho_theta and rho_q advance in place, U0 is filled when first != 0.
extern "C" int bz_tendencies_fused_rk(bz_ctx *ctx, const bz_state *s, const bz_prognostic *U0, const bz_prognostic *G,
                                      double dt, double alpha, int first)
{
    if (!ctx || !s || !U0 || !G) return BZ_ERR_INVALID;
    const DevGrid &g = ctx->dg;
    if (first) {      // wall faces of the predictor stay 0
        BZ_HIP(hipMemsetAsync(G->rho_w + g.Sxy * g.Hz, 0, g.Sxy * sizeof(double), ctx->stream));
        BZ_HIP(hipMemsetAsync(G->rho_w + g.Sxy * (g.Hz + g.Nz), 0, g.Sxy * sizeof(double), ctx->stream));
    }
    int rc = bzi_tendencies_fused_rk(ctx, s, U0, G, dt, alpha, first != 0);
    ctx->G_is_predictor = true;
    return rc;
}

extern "C" int bz_poisson_source_term_from(bz_ctx *ctx, const bz_state *s, const bz_prognostic *predictor, double dt,
                                           double *rhs)
{
    if (!ctx || !s || !predictor || !rhs) return BZ_ERR_INVALID;
    return bzi_poisson_source_fused(ctx, s, dt, rhs, predictor);
}

extern "C" int bz_project_and_diagnose_from(bz_ctx *ctx, const bz_state *s, const bz_prognostic *predictor,
                                            const double *phi_c, const double *phi_below, double dt)
{
    if (!ctx || !s || !predictor || !phi_c) return BZ_ERR_INVALID;
    if (!ctx->dg.wrap_y && !phi_below) return BZ_ERR_INVALID;
    return bzi_project_diagnose(ctx, s, dt, phi_c, phi_below, predictor);
}
