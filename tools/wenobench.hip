// wenobench.hip — pure-arithmetic cost of one WENO-5 reconstruction on MI355X (no memory traffic).
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../breeze.jl_amd/csrc/bz_weno.h"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int MODE>
__global__ __launch_bounds__(256) void k(double* out, int iters, double seed)
{
    double a = seed + threadIdx.x * 1e-3, b = a + 0.1, c = b + 0.13, d = c - 0.07, e = d + 0.21, f = e * 0.9;
    double acc = 0.0;
    for (int it = 0; it < iters; ++it) {
        double r;
        if (MODE == 0) r = bz_weno5_fast(a, b, c, d, e);
        else if (MODE == 1) r = bz_weno5_ref(a, b, c, d, e);
        else if (MODE == 2) r = bz_up5(a, b, c, d, e, f, acc > 0.5);
        else r = fma(a, b, c) * d + e;          // 2 FMA-class ops: issue-rate calibration
        acc += r;
        a = b; b = c; c = d; d = e; e = f; f = r * 0.999 + 0.001;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
int main() {
    double* out; CK(hipMalloc(&out, 256 * 256 * 8 * 5 * sizeof(double)));
    const int iters = 4000;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run = [&](const char* name, auto kern, int blocks_per_cu) {
        int blocks = 256 * blocks_per_cu;
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0); CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        // waves per SIMD = blocks_per_cu (256 threads = 4 waves = 1 per SIMD)
        double cyc = ms * 1e-3 * 2.4e9 / ((double)iters * blocks_per_cu);
        printf("%-28s waves/SIMD=%d  %8.3f ms  -> %7.1f cycles@2.4GHz per call per wave (SIMD time)\n", name, blocks_per_cu, ms, cyc);
    };
    for (int w : {1, 2, 4}) {
        run("weno5 one-division", k<0>, w);
        run("weno5 reference order", k<1>, w);
        run("up5 (select + weno5)", k<2>, w);
        run("2 fma calibration", k<3>, w);
    }
    return 0;
}
