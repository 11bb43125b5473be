// tendbench.hip — A/B timing of tendency-kernel variants on random data (standalone, not shipped).
// build: see tools/build_tendbench.sh
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
#include "../breeze.jl_amd/csrc/bz_tendency.hip"   // gen-1 kernels (also provides bz_compute_tendencies)
#include "../breeze.jl_amd/csrc/bz_tendency3_kernels.h"
#include "../breeze.jl_amd/csrc/bz_tendency4_kernels.h"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// memory-structure probes: gen-1 scalar marching loop with (a) only the streaming loads (ring + velocities),
// (b) plus the 6 x-neighbours, (c) plus the 6 y-neighbours; arithmetic is a trivial sum.
template <int MODE>
__global__ __launch_bounds__(256) void k_probe(DevGrid g, double *__restrict__ Gc, const double *__restrict__ u,
                                               const double *__restrict__ v, const double *__restrict__ w,
                                               const double *__restrict__ c, int kchunk)
{
    const int i = blockIdx.x * 64 + threadIdx.x, j = blockIdx.y * 4 + threadIdx.y;
    if (i >= g.Nx || j >= g.Ny) return;
    const int k0 = blockIdx.z * kchunk, k1 = min(k0 + kchunk, g.Nz);
    const long long sy = g.Sx, sz = g.Sxy;
    long long n = g.idx(i, j, k0);
    double r0 = c[n - 3 * sz], r1 = c[n - 2 * sz], r2 = c[n - sz], r3 = c[n], r4 = c[n + sz], r5 = c[n + 2 * sz];
    for (int k = k0; k < k1; ++k, n += sz) {
        double t = c[n + 3 * sz];
        double acc = (r0 + r1 + r2 + r3 + r4 + r5 + t) * w[n + sz] + u[n] * u[n + 1] + v[n] * v[n + sy];
        if (MODE >= 1) acc += c[n - 3] + c[n - 2] + c[n - 1] + c[n + 1] + c[n + 2] + c[n + 3];
        if (MODE >= 2) acc += c[n - 3 * sy] + c[n - 2 * sy] + c[n - sy] + c[n + sy] + c[n + 2 * sy] + c[n + 3 * sy];
        Gc[n] = acc;
        r0 = r1; r1 = r2; r2 = r3; r3 = r4; r4 = r5; r5 = t;
    }
}

template <class K> float timeit(K launch, int reps = 5) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    launch(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int r = 0; r < reps; ++r) launch();
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms / reps;
}

int main(int argc, char** argv) {
    int Nz = argc > 1 ? atoi(argv[1]) : 256, Nx = argc > 2 ? atoi(argv[2]) : 512, Ny = argc > 3 ? atoi(argv[3]) : 512, H = 3;
    std::vector<double> zf(Nz + 1); for (int k = 0; k <= Nz; ++k) zf[k] = 10e3 * k / Nz;
    bz_grid g{Nx, Ny, Nz, H, H, H, {0, 0, 1}, 8, 20e3 / Nx, 20e3 / Ny, zf.data(), 1, 0};
    bz_constants c{9.81, 8.314462618 / 0.02897, 8.314462618 / 0.018015, 1005, 1850};
    std::vector<double> rho(Nz + 2 * H), p(Nz + 2 * H), T(Nz + 2 * H);
    for (int k = 0; k < Nz + 2 * H; ++k) { rho[k] = 1.2 * exp(-(k - H) * 10e3 / Nz / 8e3); p[k] = 1e5 * exp(-(k - H) * 10e3 / Nz / 8e3); T[k] = 300 - 0.006 * (k - H) * 10e3 / Nz; }
    bz_reference_state r{101325, 300, 1e5, rho.data(), p.data(), T.data()};
    bz_ctx* ctx; int rc = bz_create(&ctx, &g, &c, &r, 5); if (rc) { printf("bz_create %d\n", rc); return 1; }
    DevGrid dg = ctx->dg;
    size_t nc = (size_t)dg.Sxy * (Nz + 2 * H + 1);
    std::vector<double> h(nc);
    auto mk = [&](double base, double amp, int seed) {
        double* d; CK(hipMalloc(&d, nc * sizeof(double)));
        srand(seed);
        for (size_t n = 0; n < nc; ++n) { size_t i = n % dg.Sx, j = (n / dg.Sx) % dg.Sy, k = n / dg.Sxy;
            h[n] = base + amp * (0.7 * sin(0.05 * i + 0.3) * cos(0.04 * j) * sin(0.03 * k + 0.1) + 0.3 * (rand() / (double)RAND_MAX - 0.5)); }
        CK(hipMemcpy(d, h.data(), nc * sizeof(double), hipMemcpyHostToDevice)); return d; };
    double *ru = mk(0, 5, 1), *rv = mk(0, 5, 2), *rw = mk(0, 2, 3), *u = mk(0, 5, 4), *v = mk(0, 5, 5), *w = mk(0, 2, 6);
    double *th = mk(300, 3, 7), *q = mk(5e-3, 2e-3, 8), *TT = mk(280, 5, 9), *G0 = mk(0, 0, 10), *G1 = mk(0, 0, 11);
    double cells = (double)Nx * Ny * Nz;
    auto rep = [&](const char* name, float ms) { printf("%-52s %8.3f ms  %7.1f ps/cell\n", name, ms, ms * 1e9 / cells); fflush(stdout); };
    Tend3Fields F3; F3.ru = ru; F3.rv = rv; F3.rw = rw; F3.u = u; F3.v = v; F3.w = w; F3.T = TT; F3.q = q; F3.G = G0;
#define RUN3(KIND, CPTR, RR, TYWV, KC) F3.c = CPTR; rep("gen3 " #KIND " R=" #RR " TYW=" #TYWV " kc=" #KC, timeit([&] { \
        dim3 grid((Nx + 63) / 64, (Ny + RR * TYWV - 1) / (RR * TYWV), ((KIND == T3_W ? Nz - 1 : Nz) + KC - 1) / KC), block(64, TYWV); \
        hipLaunchKernelGGL((k_tend3<KIND, RR, TYWV>), grid, block, 0, 0, dg, F3, KC); }))
    {
        dim3 b1(64, 4), g1((Nx + 63) / 64, (Ny + 3) / 4, 4); int kc1 = (Nz + 3) / 4;
        rep("probe: ring + velocities only", timeit([&] { hipLaunchKernelGGL(k_probe<0>, g1, b1, 0, 0, dg, G0, u, v, w, th, kc1); }));
        rep("probe: + 6 x-neighbours", timeit([&] { hipLaunchKernelGGL(k_probe<1>, g1, b1, 0, 0, dg, G0, u, v, w, th, kc1); }));
        rep("probe: + 6 x + 6 y neighbours", timeit([&] { hipLaunchKernelGGL(k_probe<2>, g1, b1, 0, 0, dg, G0, u, v, w, th, kc1); }));
        dim3 g2((Nx + 63) / 64, (Ny + 3) / 4, 16); int kc2 = (Nz + 15) / 16;
        rep("probe: ring + velocities only, 16 chunks", timeit([&] { hipLaunchKernelGGL(k_probe<0>, g2, b1, 0, 0, dg, G0, u, v, w, th, kc2); }));
        rep("probe: + 6 x + 6 y neighbours, 16 chunks", timeit([&] { hipLaunchKernelGGL(k_probe<2>, g2, b1, 0, 0, dg, G0, u, v, w, th, kc2); }));
    }
    {
        dim3 b1(64, 4), g1((Nx + 63) / 64, (Ny + 3) / 4, 4); int kc1 = (Nz + 3) / 4;
        rep("gen1 scalar (theta)", timeit([&] { hipLaunchKernelGGL(k_scalar_tendency, g1, b1, 0, 0, dg, G0, u, v, w, th, kc1); }));
        rep("gen1 u", timeit([&] { hipLaunchKernelGGL(k_u_tendency, g1, b1, 0, 0, dg, G0, ru, rv, rw, u, kc1, RKEpilogue()); }));
        rep("gen1 v", timeit([&] { hipLaunchKernelGGL(k_v_tendency, g1, b1, 0, 0, dg, G0, ru, rv, rw, v, kc1, RKEpilogue()); }));
        rep("gen1 w", timeit([&] { hipLaunchKernelGGL(k_w_tendency, g1, b1, 0, 0, dg, G0, ru, rv, rw, w, TT, q, kc1); }));
    }
    {   // fused scalar pair: gen-3 (loads) vs gen-4 (LDS tile), with a result comparison
        auto cmp = [&](const char* what) {
            std::vector<double> x(nc), y(nc);
            CK(hipMemcpy(x.data(), G0, nc * sizeof(double), hipMemcpyDeviceToHost)); CK(hipMemcpy(y.data(), G1, nc * sizeof(double), hipMemcpyDeviceToHost));
            return std::make_pair(x, y); };
        dim3 b3(64, 4), g3((Nx + 63) / 64, (Ny + 3) / 4, (Nz + 63) / 64);
        rep("pair gen3 (k_scalar_pair<4>) kc=64", timeit([&] { hipLaunchKernelGGL((k_scalar_pair<4>), g3, b3, 0, 0, dg, u, v, w, th, q, G0, G1, 64); }));
        auto ref = cmp("ref");
        CK(hipMemset(G0, 0, nc * sizeof(double))); CK(hipMemset(G1, 0, nc * sizeof(double)));
        dim3 b8(64, 8), g8((Nx + 63) / 64, (Ny + 7) / 8, (Nz + 63) / 64);
        rep("pair gen4 LDS TY=8 kc=64", timeit([&] { hipLaunchKernelGGL((k_scalar_pair_lds<8>), g8, b8, 0, 0, dg, u, v, w, th, q, G0, G1, 64, RKEpilogue(), th, q); }));
        auto got = cmp("got");
        double e0 = 0, e1 = 0, s0 = 0, s1 = 0;
        for (size_t n = 0; n < nc; ++n) { e0 = fmax(e0, fabs(got.first[n] - ref.first[n])); e1 = fmax(e1, fabs(got.second[n] - ref.second[n])); s0 = fmax(s0, fabs(ref.first[n])); s1 = fmax(s1, fabs(ref.second[n])); }
        printf("   gen4 vs gen3: max|dG_theta| = %.3e (scale %.3e), max|dG_q| = %.3e (scale %.3e)\n", e0, s0, e1, s1);
        rep("pair gen4 LDS TY=4 kc=128", timeit([&] { hipLaunchKernelGGL((k_scalar_pair_lds<4>), dim3((Nx + 63) / 64, (Ny + 3) / 4, (Nz + 127) / 128), dim3(64, 4), 0, 0, dg, u, v, w, th, q, G0, G1, 128, RKEpilogue(), th, q); }));
        rep("pair gen4 LDS TY=4 kc=256", timeit([&] { hipLaunchKernelGGL((k_scalar_pair_lds<4>), dim3((Nx + 63) / 64, (Ny + 3) / 4, (Nz + 255) / 256), dim3(64, 4), 0, 0, dg, u, v, w, th, q, G0, G1, 256, RKEpilogue(), th, q); }));
        dim3 g8b((Nx + 63) / 64, (Ny + 7) / 8, (Nz + 127) / 128);
        rep("pair gen4 LDS TY=8 kc=128", timeit([&] { hipLaunchKernelGGL((k_scalar_pair_lds<8>), g8b, b8, 0, 0, dg, u, v, w, th, q, G0, G1, 128, RKEpilogue(), th, q); }));
        dim3 b4(64, 4), g4((Nx + 63) / 64, (Ny + 3) / 4, (Nz + 63) / 64);
        rep("pair gen4 LDS TY=4 kc=64", timeit([&] { hipLaunchKernelGGL((k_scalar_pair_lds<4>), g4, b4, 0, 0, dg, u, v, w, th, q, G0, G1, 64, RKEpilogue(), th, q); }));
    }
    {   // momentum: gen-1 / ring kernels vs the LDS y-tile kernels, with result comparison
        auto grab = [&](double* p) { std::vector<double> x(nc); CK(hipMemcpy(x.data(), p, nc * sizeof(double), hipMemcpyDeviceToHost)); return x; };
        auto diff = [&](const std::vector<double>& a, const std::vector<double>& b) { double e = 0, s = 0; for (size_t n = 0; n < nc; ++n) { e = fmax(e, fabs(a[n] - b[n])); s = fmax(s, fabs(b[n])); } printf("   max|diff| = %.3e (scale %.3e)\n", e, s); };
        dim3 b1(64, 4), g1((Nx + 63) / 64, (Ny + 3) / 4, 4); int kc1 = (Nz + 3) / 4;
        Tend3Fields Fm; Fm.ru = ru; Fm.rv = rv; Fm.rw = rw; Fm.u = u; Fm.v = v; Fm.w = w; Fm.T = TT; Fm.q = q; Fm.G = G0;
        CK(hipMemset(G0, 0, nc * sizeof(double)));
        rep("u gen1", timeit([&] { hipLaunchKernelGGL(k_u_tendency, g1, b1, 0, 0, dg, G0, ru, rv, rw, u, kc1, RKEpilogue()); }));
        auto uref = grab(G0); CK(hipMemset(G0, 0, nc * sizeof(double)));
        Fm.c = u;
        rep("u LDS TY=8 kc=64", timeit([&] { hipLaunchKernelGGL((k_u_tend_lds<8>), dim3((Nx + 63) / 64, (Ny + 7) / 8, (Nz + 63) / 64), dim3(64, 8), 0, 0, dg, Fm, 64, RKEpilogue()); }));
        diff(grab(G0), uref);
        rep("u LDS TY=4 kc=64", timeit([&] { hipLaunchKernelGGL((k_u_tend_lds<4>), dim3((Nx + 63) / 64, (Ny + 3) / 4, (Nz + 63) / 64), dim3(64, 4), 0, 0, dg, Fm, 64, RKEpilogue()); }));
        rep("u LDS TY=8 kc=256", timeit([&] { hipLaunchKernelGGL((k_u_tend_lds<8>), dim3((Nx + 63) / 64, (Ny + 7) / 8, (Nz + 255) / 256), dim3(64, 8), 0, 0, dg, Fm, 256, RKEpilogue()); }));
        rep("u LDS TY=4 kc=128", timeit([&] { hipLaunchKernelGGL((k_u_tend_lds<4>), dim3((Nx + 63) / 64, (Ny + 3) / 4, (Nz + 127) / 128), dim3(64, 4), 0, 0, dg, Fm, 128, RKEpilogue()); }));
        rep("u LDS TY=4 kc=256", timeit([&] { hipLaunchKernelGGL((k_u_tend_lds<4>), dim3((Nx + 63) / 64, (Ny + 3) / 4, (Nz + 255) / 256), dim3(64, 4), 0, 0, dg, Fm, 256, RKEpilogue()); }));
        rep("u LDS TY=8 kc=128", timeit([&] { hipLaunchKernelGGL((k_u_tend_lds<8>), dim3((Nx + 63) / 64, (Ny + 7) / 8, (Nz + 127) / 128), dim3(64, 8), 0, 0, dg, Fm, 128, RKEpilogue()); }));
        CK(hipMemset(G0, 0, nc * sizeof(double)));
        rep("v gen1", timeit([&] { hipLaunchKernelGGL(k_v_tendency, g1, b1, 0, 0, dg, G0, ru, rv, rw, v, kc1, RKEpilogue()); }));
        auto vref = grab(G0); CK(hipMemset(G0, 0, nc * sizeof(double)));
        Fm.c = v;
        rep("v LDS TY=8 kc=128", timeit([&] { hipLaunchKernelGGL((k_v_tend_lds<8>), dim3((Nx + 63) / 64, (Ny + 7) / 8, (Nz + 127) / 128), dim3(64, 8), 0, 0, dg, Fm, 128, RKEpilogue()); }));
        printf("v LDS vs gen1:"); diff(grab(G0), vref);
        rep("v LDS TY=8 kc=64", timeit([&] { hipLaunchKernelGGL((k_v_tend_lds<8>), dim3((Nx + 63) / 64, (Ny + 7) / 8, (Nz + 63) / 64), dim3(64, 8), 0, 0, dg, Fm, 64, RKEpilogue()); }));
        CK(hipMemset(G0, 0, nc * sizeof(double)));
        Fm.c = w;
        rep("w ring TYW=4 kc=64", timeit([&] { hipLaunchKernelGGL((k_w_tend_ring<4>), dim3((Nx + 63) / 64, (Ny + 3) / 4, (Nz - 1 + 63) / 64), dim3(64, 4), 0, 0, dg, Fm, 64, RKEpilogue()); }));
        auto wref = grab(G0); CK(hipMemset(G0, 0, nc * sizeof(double)));
        rep("w LDS TY=8 kc=64", timeit([&] { hipLaunchKernelGGL((k_w_tend_lds<8>), dim3((Nx + 63) / 64, (Ny + 7) / 8, (Nz - 1 + 63) / 64), dim3(64, 8), 0, 0, dg, Fm, 64, RKEpilogue()); }));
        diff(grab(G0), wref);
        rep("w LDS TY=8 kc=128", timeit([&] { hipLaunchKernelGGL((k_w_tend_lds<8>), dim3((Nx + 63) / 64, (Ny + 7) / 8, (Nz - 1 + 127) / 128), dim3(64, 8), 0, 0, dg, Fm, 128, RKEpilogue()); }));
        rep("w LDS TY=8 kc=256", timeit([&] { hipLaunchKernelGGL((k_w_tend_lds<8>), dim3((Nx + 63) / 64, (Ny + 7) / 8, (Nz - 1 + 255) / 256), dim3(64, 8), 0, 0, dg, Fm, 256, RKEpilogue()); }));
        rep("w ring TYW=4 kc=128", timeit([&] { hipLaunchKernelGGL((k_w_tend_ring<4>), dim3((Nx + 63) / 64, (Ny + 3) / 4, (Nz - 1 + 127) / 128), dim3(64, 4), 0, 0, dg, Fm, 128, RKEpilogue()); }));
        CK(hipMemset(G0, 0, nc * sizeof(double)));
        hipLaunchKernelGGL((k_w_tend_lds<4>), dim3((Nx + 63) / 64, (Ny + 3) / 4, (Nz - 1 + 63) / 64), dim3(64, 4), 0, 0, dg, Fm, 64, RKEpilogue());
        printf("w LDS TY=4 vs ring:"); diff(grab(G0), wref);
        rep("w LDS TY=4 kc=64", timeit([&] { hipLaunchKernelGGL((k_w_tend_lds<4>), dim3((Nx + 63) / 64, (Ny + 3) / 4, (Nz - 1 + 63) / 64), dim3(64, 4), 0, 0, dg, Fm, 64, RKEpilogue()); }));
    }
    RUN3(T3_SCALAR, th, 1, 4, 64);
    RUN3(T3_SCALAR, th, 2, 4, 64);
    RUN3(T3_SCALAR, th, 2, 4, 128);
    RUN3(T3_SCALAR, th, 2, 2, 64);
    RUN3(T3_SCALAR, th, 3, 4, 64);
    RUN3(T3_SCALAR, th, 4, 4, 64);
    RUN3(T3_U, u, 1, 4, 64);
    RUN3(T3_U, u, 2, 4, 64);
    RUN3(T3_U, u, 3, 4, 64);
    RUN3(T3_V, v, 1, 4, 64);
    RUN3(T3_V, v, 2, 4, 64);
    RUN3(T3_W, w, 1, 4, 64);
    RUN3(T3_W, w, 2, 4, 64);
    RUN3(T3_W, w, 3, 4, 64);
    return 0;
}
