// tools/mall_probe.hip — does the 256 MB Infinity Cache keep a chunk of the Poisson spectrum on the die between the three middle passes
// (y transform, tridiagonal solve, inverse y transform: each reads and writes the whole 1.08 GB half spectrum of 512^3)?  Three in-place
// streaming kernels over a 1.08 GB buffer, (a) each over the whole buffer, (b) chunk by chunk — the three passes of a chunk back to back —
// for chunk sizes 16 ... 256 MB.  Prints ms per triple pass and the bytes per second it amounts to.   VERDICT r05 item 5.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(256) void k_touch(double2 *p, long long n, double a)
{
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        double2 v = p[i];
        v.x = v.x * a + v.y; v.y = v.y * a - v.x;
        p[i] = v;
    }
}
int main()
{
    const long long n = 257LL * 512 * 512;      // double2 entries: 1.08 GB
    double2 *d;
    hipMalloc(&d, n * sizeof(double2));
    hipMemset(d, 0, n * sizeof(double2));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](long long chunk_bytes) {
        const long long ce = chunk_bytes / (long long)sizeof(double2);
        float best = 1e9f;
        for (int rep = 0; rep < 5; ++rep) {
            hipEventRecord(e0);
            for (long long o = 0; o < n; o += ce) {
                const long long m = (n - o < ce) ? n - o : ce;
                const int blocks = (int)((m / 256 / 4 < 1) ? 1 : (m / 256 / 4 > 65535 * 4 ? 65535 * 4 : m / 256 / 4));
                for (int pass = 0; pass < 3; ++pass) hipLaunchKernelGGL(k_touch, dim3(blocks), dim3(256), 0, 0, d + o, m, 1.0 + 1e-9 * pass);
            }
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        printf("chunk %7.1f MB: %.3f ms per triple pass = %.2f TB/s of read + written bytes\n", chunk_bytes / 1048576.0, best,
               6.0 * n * sizeof(double2) / best / 1e9);
    };
    run(n * sizeof(double2));
    for (long long mb : {256, 192, 128, 96, 64, 32, 16}) run(mb << 20);
    return 0;
}
