// Strided 128-byte-piece copy: does the order of the spectral array (level-major vs kx-major) matter to HBM?
//   hipcc -O3 --offload-arch=gfx950 tools/stride_bench.hip -o tools/stride_bench && tools/stride_bench
// A workgroup of 512 threads owns 8 adjacent columns (one 128-byte line per level) and 64 segments of 8 levels, like k_tridiag_coop;
// it reads its 8 x 512 double2, scales and writes them back.  Layout 0: hat[k][kx][ky] (level stride = NXH * Ny elements);
// layout 1: hat[kx][k][ky] (level stride = Ny, kx stride = Nz * Ny).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
template <int COLS>
__global__ __launch_bounds__(512) void k_pieces(double2 *hat, long long level_stride, long long group_stride, int groups_per_line, long long line_stride)
{
    constexpr int SEGS = 512 / COLS, M = 512 / SEGS;
    const int t = threadIdx.x, cc = t % COLS, s = t / COLS;
    const long long grp = blockIdx.x;
    double2 *col = hat + (grp / groups_per_line) * line_stride + (grp % groups_per_line) * group_stride + cc;
    double2 v[M];
#pragma unroll
    for (int j = 0; j < M; ++j) v[j] = col[level_stride * (s * M + j)];
#pragma unroll
    for (int j = 0; j < M; ++j) { v[j].x *= 1.0000001; v[j].y *= 0.9999999; }
#pragma unroll
    for (int j = 0; j < M; ++j) col[level_stride * (s * M + j)] = v[j];
}
__global__ __launch_bounds__(512) void k_contig(double2 *hat)
{
    double2 *col = hat + (long long)blockIdx.x * 4096 + threadIdx.x;
    double2 v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = col[512 * j];
#pragma unroll
    for (int j = 0; j < 8; ++j) { v[j].x *= 1.0000001; v[j].y *= 0.9999999; }
#pragma unroll
    for (int j = 0; j < 8; ++j) col[512 * j] = v[j];
}
int main()
{
    const int NXH = 257, Ny = 512, Nz = 512;
    const long long n = (long long)NXH * Ny * Nz;
    double2 *hat;
    hipMalloc(&hat, n * sizeof(double2));
    hipMemset(hat, 0, n * sizeof(double2));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](const char *name, auto launch) {
        for (int w = 0; w < 3; ++w) launch();
        hipEventRecord(e0);
        for (int r = 0; r < 20; ++r) launch();
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 20;
        printf("%-46s %.3f ms  %.2f TB/s\n", name, ms, 2.0 * n * 16 / ms * 1e-9);
    };
    const long long plane = (long long)NXH * Ny;
    run("level-major, 8 cols (128 B pieces)", [&] { hipLaunchKernelGGL(k_pieces<8>, dim3(plane / 8), dim3(512), 0, 0, hat, plane, 8LL, (int)(plane / 8), 0LL); });
    run("kx-major,    8 cols (128 B pieces)", [&] { hipLaunchKernelGGL(k_pieces<8>, dim3(plane / 8), dim3(512), 0, 0, hat, (long long)Ny, 8LL, Ny / 8, (long long)Nz * Ny); });
    run("level-major, 16 cols (256 B pieces)", [&] { hipLaunchKernelGGL(k_pieces<16>, dim3(plane / 16), dim3(512), 0, 0, hat, plane, 16LL, (int)(plane / 16), 0LL); });
    run("kx-major,    16 cols (256 B pieces)", [&] { hipLaunchKernelGGL(k_pieces<16>, dim3(plane / 16), dim3(512), 0, 0, hat, (long long)Ny, 16LL, Ny / 16, (long long)Nz * Ny); });
    run("level-major, 32 cols (512 B pieces)", [&] { hipLaunchKernelGGL(k_pieces<32>, dim3(plane / 32), dim3(512), 0, 0, hat, plane, 32LL, (int)(plane / 32), 0LL); });
    run("kx-major,    32 cols (512 B pieces)", [&] { hipLaunchKernelGGL(k_pieces<32>, dim3(plane / 32), dim3(512), 0, 0, hat, (long long)Ny, 32LL, Ny / 32, (long long)Nz * Ny); });
    run("contiguous (512 threads x 8 double2)", [&] { hipLaunchKernelGGL(k_contig, dim3(plane / 8), dim3(512), 0, 0, hat); });
    return 0;
}
