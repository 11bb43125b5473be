// tools/fused_sweep_probe.hip — could ONE kernel per acoustic substep (forward sweep + back substitution + recovery, the predictors and the
// eliminated right-hand side never leaving the chip: 26 words per cell instead of 32) run faster than today's pair (2.15 + 0.9 ms at
// 512 x 512 x 256)?  The structure such a kernel needs, with placeholder arithmetic: a block owns 64 columns x Nz levels; wave w of NW
// takes the levels [w Nz / NW, (w + 1) Nz / NW) — its pointwise work (NR own loads + NY loads of the rows j - 1 / j + 1 per level), two
// values per level kept in registers (the predictors), one in LDS (the eliminated right-hand side, 64 x Nz x 8 B); the first-order
// recurrences run per chunk and are stitched through LDS (chunk summaries + fix-up); then the backward pass over the same chunk stores NS
// arrays.  Prints ms per launch and the bytes per second of the (NR + NS) compulsory words.  VERDICT r05 item 1 (the fused form).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

constexpr int NRMAX = 16, NSMAX = 10;
struct Arrays {
    const double *in[NRMAX];
    double *out[NSMAX];
};

template <int NW, int LEV, int NR, int NY, int NS>
__global__ __launch_bounds__(64 * NW) void k_fused(Arrays A, int Nx, int Ny, long long sx, long long sxy, double c)
{
    extern __shared__ double lds[];      // phi[Nz][64], then summaries[NW][64] x 2
    const int lane = threadIdx.x, w = threadIdx.y;
    const int i = blockIdx.x * 64 + lane, j = blockIdx.y;
    const int Nz = NW * LEV;
    double *phi = lds, *sumA = lds + (size_t)Nz * 64, *sumB = sumA + NW * 64;
    const long long base = (long long)i + sx * j;
    const long long jm = (j > 0) ? -sx : sx * (Ny - 1), jp = (j + 1 < Ny) ? sx : -sx * (Ny - 1);
    double r1[LEV], r2[LEV];
    double carry = 0.0;
#pragma unroll
    for (int l = 0; l < LEV; ++l) {
        const int k = w * LEV + l;
        const long long n = base + sxy * k;
        double v[NR];
#pragma unroll
        for (int a = 0; a < NR; ++a) v[a] = A.in[a][n];
        double y = 0.0;
#pragma unroll
        for (int a = 0; a < NY; ++a) y += A.in[a][n + ((a & 1) ? jp : jm)];
        double s1 = v[0], s2 = v[1];
#pragma unroll
        for (int a = 2; a < NR; ++a) { if (a & 1) s1 = fma(v[a], c, s1); else s2 = fma(v[a], c, s2); }
        s1 = fma(y, c, s1);
        r1[l] = s1; r2[l] = s2;
        carry = fma(carry, c, s1 + s2);      // the forward recurrence of the chunk (started from zero)
        phi[(size_t)k * 64 + lane] = carry;
    }
    sumA[w * 64 + lane] = carry;
    __syncthreads();
    // stitch: what the chunks below contribute to this chunk's start (every wave walks the summaries below it)
    double pre = 0.0;
    for (int q = 0; q < w; ++q) pre = fma(pre, c, sumA[q * 64 + lane]);
    // backward recurrence of the chunk from zero + summary
    double wk = 0.0, cp = 1.0;
#pragma unroll 4
    for (int l = LEV - 1; l >= 0; --l) {
        const int k = w * LEV + l;
        wk = fma(wk, c, phi[(size_t)k * 64 + lane] + pre * cp);
        cp *= c;
    }
    sumB[w * 64 + lane] = wk;
    __syncthreads();
    double post = 0.0;
    for (int q = NW - 1; q > w; --q) post = fma(post, c, sumB[q * 64 + lane]);
    wk = post;
#pragma unroll
    for (int l = LEV - 1; l >= 0; --l) {
        const int k = w * LEV + l;
        const long long n = base + sxy * k;
        const double wn = fma(wk, c, phi[(size_t)k * 64 + lane] + pre);
#pragma unroll
        for (int a = 0; a < NS; ++a) A.out[a][n] = fma(r1[l], (double)(a + 1), r2[l]) + ((a & 1) ? wn : wk);
        wk = wn;
    }
}

// today's structure with the same placeholder arithmetic: a thread per column marching all levels, NR loads + NS stores per level
template <int NR, int NY, int NS>
__global__ __launch_bounds__(256) void k_march(Arrays A, int Nx, int Ny, int Nz, long long sx, long long sxy, double c)
{
    const int i = blockIdx.x * 64 + threadIdx.x, j = blockIdx.y * 4 + threadIdx.y;
    const long long base = (long long)i + sx * j;
    const long long jm = (j > 0) ? -sx : sx * (Ny - 1), jp = (j + 1 < Ny) ? sx : -sx * (Ny - 1);
    double carry = 0.0;
    for (int k = 0; k < Nz; ++k) {
        const long long n = base + sxy * k;
        double v[NR];
#pragma unroll
        for (int a = 0; a < NR; ++a) v[a] = A.in[a][n];
        double y = 0.0;
#pragma unroll
        for (int a = 0; a < NY; ++a) y += A.in[a][n + ((a & 1) ? jp : jm)];
        double s1 = v[0], s2 = v[1];
#pragma unroll
        for (int a = 2; a < NR; ++a) { if (a & 1) s1 = fma(v[a], c, s1); else s2 = fma(v[a], c, s2); }
        s1 = fma(y, c, s1);
        carry = fma(carry, c, s1 + s2);
#pragma unroll
        for (int a = 0; a < NS; ++a) A.out[a][n] = fma(s1, (double)(a + 1), s2) + carry;
    }
}

int main()
{
    const int Nx = 512, Ny = 512, Nz = 256;
    const long long sx = Nx, sxy = (long long)Nx * Ny, n = sxy * Nz;
    Arrays A;
    for (int a = 0; a < NRMAX; ++a) { double *p; if (hipMalloc(&p, n * 8) != hipSuccess) { printf("alloc\n"); return 1; } hipMemset(p, 0, n * 8); A.in[a] = p; }
    for (int a = 0; a < NSMAX; ++a) { double *p; if (hipMalloc(&p, n * 8) != hipSuccess) { printf("alloc\n"); return 1; } hipMemset(p, 0, n * 8); A.out[a] = p; }
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    auto time = [&](const char *name, int words, auto launch) {
        float best = 1e9f;
        for (int rep = 0; rep < 6; ++rep) {
            hipEventRecord(e0);
            launch();
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (rep && ms < best) best = ms;
        }
        hipError_t err = hipGetLastError();
        printf("%-58s %.3f ms  = %.2f TB/s of %d words%s\n", name, best, words * 8.0 * n / best / 1e9, words, err == hipSuccess ? "" : "  LAUNCH ERROR");
    };
#define FUSED(NW, NR, NY, NS)                                                                                                  \
    {                                                                                                                            \
        constexpr int LEV = 256 / NW;                                                                                            \
        const size_t lds = ((size_t)Nz * 64 + 2 * NW * 64) * 8;                                                                  \
        hipFuncSetAttribute((const void *)k_fused<NW, LEV, NR, NY, NS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);  \
        char nm[96]; snprintf(nm, sizeof nm, "fused: %2d waves x %3d levels, %2d + %d loads, %2d stores", NW, LEV, NR, NY, NS);     \
        time(nm, NR + NS, [&] { hipLaunchKernelGGL((k_fused<NW, LEV, NR, NY, NS>), dim3(Nx / 64, Ny), dim3(64, NW), lds, 0, A, Nx, Ny, sx, sxy, 0.5); }); \
    }
    FUSED(8, 16, 8, 10)
    FUSED(16, 16, 8, 10)
    FUSED(8, 16, 0, 10)
    FUSED(16, 16, 0, 10)
#define MARCH(NR, NY, NS)                                                                                                      \
    {                                                                                                                            \
        char nm[96]; snprintf(nm, sizeof nm, "march (a column per thread, 64 x 4 blocks): %2d + %d loads, %2d stores", NR, NY, NS); \
        time(nm, NR + NS, [&] { hipLaunchKernelGGL((k_march<NR, NY, NS>), dim3(Nx / 64, Ny / 4), dim3(64, 4), 0, 0, A, Nx, Ny, Nz, sx, sxy, 0.5); }); \
    }
    MARCH(16, 8, 10)
    MARCH(15, 8, 7)      // today's forward sweep: 22 words
    MARCH(6, 0, 4)       // today's backward sweep: 10 words
    return 0;
}
