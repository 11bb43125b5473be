// bz_halo.hip — fill_halo_regions! for (Periodic, Periodic, Bounded) fields.
// Semantics (Oceananigans.BoundaryConditions, call sites
// /root/reference/src/AtmosphereModels/update_atmosphere_model_state.jl:48,135-136,152,241-243):
//   Periodic x,y : wrap copy of all H halo cells;
//   Bounded z    : kind 0 centre field, default no-flux BC -> first halo cell = adjacent interior cell
//                  kind 1 z-face field, impenetrable walls  -> wall faces k=0 and k=Nz set to 0
//                  kind 2/3 `nothing` BC (diagnostic velocities) -> untouched in z.
//   Bounded y    : (topology (Periodic, Bounded, Bounded)) the same two conventions by row: a field that is a centre in y takes its
//                  first halo row from the adjacent interior row; a y-face field (kind | BZ_HALO_YFACE: rho v, v) gets zeros on its
//                  wall faces j = 0 and j = Ny (face Ny lives in the first upper halo row).
#include "bz_internal.h"

#define BZ_MAX_HALO_FIELDS 12

struct HaloList {
    double *f[BZ_MAX_HALO_FIELDS];
    int nzt[BZ_MAX_HALO_FIELDS];   // total z levels of the parent array
    int kind[BZ_MAX_HALO_FIELDS];
};

__global__ __launch_bounds__(64) void k_halo_x(DevGrid g, HaloList L)
{
    int fi = blockIdx.z, kk = blockIdx.y;
    if (kk >= L.nzt[fi]) return;
    int j = blockIdx.x * 64 + threadIdx.x;
    if (j >= g.Ny) return;
    double *row = L.f[fi] + g.Sxy * kk + (long long)g.Sx * (j + g.Hy);
    if (g.bounded_x) {      // walls in x: no-flux cell for a centre-in-x field, zero wall faces i = 0, Nx for an x-face field
        if (L.kind[fi] & BZ_HALO_XFACE) { row[g.Hx] = 0.0; row[g.Hx + g.Nx] = 0.0; }
        else { row[g.Hx - 1] = row[g.Hx]; row[g.Hx + g.Nx] = row[g.Hx + g.Nx - 1]; }
        return;
    }
    for (int h = 0; h < g.Hx; ++h) {
        row[h] = row[h + g.Nx];
        row[g.Hx + g.Nx + h] = row[g.Hx + h];
    }
}

__global__ __launch_bounds__(256) void k_halo_y(DevGrid g, HaloList L)
{
    int fi = blockIdx.z, kk = blockIdx.y;
    if (kk >= L.nzt[fi]) return;
    int ii = blockIdx.x * 256 + threadIdx.x;
    if (ii >= g.Sx) return;
    double *pl = L.f[fi] + g.Sxy * kk + ii;
    for (int h = 0; h < g.Hy; ++h) {
        pl[(long long)g.Sx * h] = pl[(long long)g.Sx * (h + g.Ny)];
        pl[(long long)g.Sx * (g.Hy + g.Ny + h)] = pl[(long long)g.Sx * (g.Hy + h)];
    }
}

__global__ __launch_bounds__(256) void k_halo_y_bounded(DevGrid g, HaloList L)
{
    int fi = blockIdx.z, kk = blockIdx.y;
    if (kk >= L.nzt[fi]) return;
    int ii = blockIdx.x * 256 + threadIdx.x;
    if (ii >= g.Sx) return;
    double *pl = L.f[fi] + g.Sxy * kk + ii;
    const long long sy = g.Sx;
    if (L.kind[fi] & BZ_HALO_YFACE) {
        pl[sy * g.Hy] = 0.0;
        pl[sy * (g.Hy + g.Ny)] = 0.0;
    } else {
        pl[sy * (g.Hy - 1)] = pl[sy * g.Hy];
        pl[sy * (g.Hy + g.Ny)] = pl[sy * (g.Hy + g.Ny - 1)];
    }
}

__global__ __launch_bounds__(256) void k_halo_z(DevGrid g, HaloList L)
{
    int fi = blockIdx.z;
    int kind = L.kind[fi] & 3;
    if (kind >= 2) return;
    long long n = (long long)blockIdx.x * 256 + threadIdx.x;
    if (n >= g.Sxy) return;
    double *f = L.f[fi];
    if (kind == 0) {
        f[g.Sxy * (g.Hz - 1) + n] = f[g.Sxy * g.Hz + n];
        f[g.Sxy * (g.Hz + g.Nz) + n] = f[g.Sxy * (g.Hz + g.Nz - 1) + n];
    } else {
        f[g.Sxy * g.Hz + n] = 0.0;
        f[g.Sxy * (g.Hz + g.Nz) + n] = 0.0;
    }
}

int bzi_fill_halos_multi(bz_ctx *ctx, double *const *fields, const int *kinds, int n)
{
    if (n <= 0) return BZ_OK;
    if (n > BZ_MAX_HALO_FIELDS) return BZ_ERR_INVALID;
    ProfileScope ps(ctx, "fill_halo_regions");
    const DevGrid &g = ctx->dg;
    HaloList L;
    int nzmax = 0;
    bool anyz = false;
    for (int i = 0; i < n; ++i) {
        if (!fields[i]) return BZ_ERR_INVALID;
        L.f[i] = fields[i];
        L.kind[i] = kinds[i];
        const int zk = kinds[i] & 3;
        L.nzt[i] = g.Nz + 2 * g.Hz + ((zk == 1 || zk == 3) ? 1 : 0);
        if (L.nzt[i] > nzmax) nzmax = L.nzt[i];
        if (zk < 2) anyz = true;
    }
    hipLaunchKernelGGL(k_halo_x, dim3((g.Ny + 63) / 64, nzmax, n), dim3(64), 0, ctx->stream, g, L);
    if (g.wrap_y)
        hipLaunchKernelGGL(k_halo_y, dim3((g.Sx + 255) / 256, nzmax, n), dim3(256), 0, ctx->stream, g, L);
    else if (g.bounded_y)      // after the x fill: whole rows, x halos included
        hipLaunchKernelGGL(k_halo_y_bounded, dim3((g.Sx + 255) / 256, nzmax, n), dim3(256), 0, ctx->stream, g, L);
    if (anyz)
        hipLaunchKernelGGL(k_halo_z, dim3((unsigned)((g.Sxy + 255) / 256), 1, n), dim3(256), 0, ctx->stream, g, L);
    BZ_LAUNCH_CHECK();
    return BZ_OK;
}

int bzi_fill_halo(bz_ctx *ctx, double *f, int kind) { return bzi_fill_halos_multi(ctx, &f, &kind, 1); }

extern "C" int bz_fill_halo_regions(bz_ctx *ctx, double *field, int kind)
{
    if (!ctx || !field || kind < 0 || kind > 15) return BZ_ERR_INVALID;
    return bzi_fill_halo(ctx, field, kind);
}
