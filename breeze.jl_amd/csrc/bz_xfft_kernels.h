// bz_xfft_kernels.h — hand-written x transforms of the pressure solve, fused with their neighbours.
//
// The library pipeline  source term -> 2-D R2C (x rows, then y columns at stride Nx/2+1) -> vertical solves -> 2-D C2R -> projection
// spends two full passes over the grid on work that needs no pass of its own, and its column transforms run at half the rate of
// the row transforms (profiles/r02_v2_rocprofv3_kernel_stats.csv: 0.72 ms against 0.38 ms per launch at 512^3).  Here
//   * k_x_forward  evaluates the source term (compute_anelastic_source_term!, /root/reference/src/AnelasticEquations/
//     anelastic_pressure_solver.jl:90-105) of RB rows into LDS, transforms the rows there (real transform of length Nx = complex
//     transform of length Nx/2 + split) and stores the half spectrum TRANSPOSED, hatT[(k * NXH + kx) * Ny + ky-index], so that
//   * the y transforms are contiguous batched 1-D plans (the library's fast case) and the vertical solves run in the ky-fastest
//     layout the slab mode already uses, and
//   * k_x_inverse  reads RB transposed rows, inverts them in LDS and stores phi for the projection kernels.  (Applying
//     make_pressure_correction! straight from LDS in the same kernel was built and measured at 512^3: 2.3 ms with two level buffers
//     in LDS, 4.0-4.4 ms with the previous level in registers — 168 VGPRs, one workgroup per CU — against 0.6 + 1.5 ms for this
//     kernel followed by k_project_lean, so it was dropped; so was requesting level k+1 into registers before transforming level k:
//     the extra registers cost a workgroup per CU and 10-20 % of the rate.  Round 3 repeated the prefetch the way that paid in
//     k_tridiag_coop — unconditional clamped loads into named registers, LDS-only barriers, first level peeled, the ISA waiting row by
//     row with the stores left in flight: 0.545 -> 0.549 ms in Float64 (100 VGPRs: two workgroups per CU instead of three),
//     0.367 -> 0.353 ms in Float32.  The loads are not what this kernel waits for; its LDS stages are.)
// Transform: Stockham autosort, radix-4 stages plus one radix-2 stage when log2(Nx/2) is odd (and a leading radix-3 stage for rows of
// 3 * 2^m cells), a team of Nx/8 threads per row,
// twiddles from a table in LDS (exp(-2 pi i t / Nx), t < 3 Nx / 4, computed on the host in the working precision).
// Included by bz_fused.hip.
#pragma once

typedef double xf_pair2 __attribute__((ext_vector_type(2), aligned(sizeof(double))));      // two consecutive reals, element-aligned
#define XF_RB 8        // rows of y per workgroup: a transposed store / load moves XF_RB consecutive complex numbers (128 B) per kx
#ifndef XF_XCD
#define XF_XCD 1
#endif

// Block order of the two x transforms.  blockIdx.x counts groups of XF_RB rows, and consecutive workgroup ids go round-robin to the 8 XCDs: in
// launch order the groups j0 and j0 + XF_RB — which in Float32 share every 128-byte line of the transposed spectrum (XF_RB complex numbers of
// 8 bytes are half a line) — sat behind different L2s, and every line crossed the fabric twice (PMC r05: k_x_inverse_f32 1.61 GB per launch for
// 1.07 GB of compulsory bytes; Float64, whose pieces are whole lines, 1.00x).  Here XCD c owns the band of row groups [c gx/8, (c+1) gx/8) of
// every level chunk and walks it in order, so the second half of a line is asked for by the next workgroup of the same XCD.
__device__ __forceinline__ void xf_block(int &bj, int &bk)
{
    bj = blockIdx.x; bk = blockIdx.y;
    const unsigned gx = gridDim.x;
    if (XF_XCD && (gx & 7u) == 0) {
        const unsigned w = blockIdx.y * gx + blockIdx.x, c = w & 7u, r = w >> 3, band = gx >> 3;
        bj = (int)(c * band + r % band);
        bk = (int)(r / band);
    }
}

// LDS slot of element p < n2 of a row: the low four bits (the 16-byte slot inside a 256-byte bank period) are XOR-swizzled with a
// function of bits 4-7, chosen by exhaustive search over linear swizzles against the access patterns of this file on the wave64
// ds_read/write_b128 lane groups (MI355X_MICROARCH.md, LDS): strided stores of the radix-4 stages (4, 16, 64 elements apart), the
// contiguous loads, the split pairs — 99 LDS passes per row transform against 192 unswizzled or padded (96 = conflict-free).
// Element n2 (the Nyquist slot of the real transform) keeps its own index; rows are n2 + 4 slots apart, which makes the transposed
// (kx, row) accesses of 8 rows conflict-free as well.
__device__ __forceinline__ int xf_slot(int p, int n2)
{
    const int h = p >> 4;
    const int x = ((h & 2) ? 6 : 0) ^ ((h & 4) ? 15 : 0) ^ ((h & 8) ? 9 : 0);
    return (p < n2) ? (p ^ x) : p;
}
#define XF_P(p) xf_slot((p), n2)
#define XF_ROW_STRIDE(n2) ((n2) + 4)

// A team of Nx / 8 <= 64 threads sits inside one wavefront and a row's transform touches that row only, so the stages need ordering
// inside the wave, not a workgroup barrier: LDS instructions of one wave execute in order; the fences keep the compiler from moving
// LDS accesses across the point.  Workgroup barriers remain where rows change hands (transposed loads / stores).
__device__ __forceinline__ void xf_wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// TW = wavefronts per team: rows of 1024 cells have teams of 128 threads, whose stages need the workgroup barrier (every team of the
// workgroup runs the same sequence of stages, so the barrier is reached uniformly)
template <int TW>
__device__ __forceinline__ void xf_sync()
{
    if (TW == 1) xf_wave_sync();
    else __syncthreads();
}

__device__ __forceinline__ double2 xf_cmul(double2 a, double2 b) { return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }

// In-place complex transform of length n2 (a power of two >= 8, or 3 * 2^m >= 48) of `row` by the team's T = n2 / 4 threads (tid = 0 .. T-1).
// Wave-level ordering only (xf_wave_sync): callers put a workgroup barrier where other waves' data is involved.
// INV: conjugated twiddles (unnormalised inverse).
// Wst: the twiddles of the radix-4 stages STAGE-MAJOR (xf_stage_twiddles): stage Ns holds w^k, w^2k, w^3k for k < Ns as three runs of
// Ns consecutive entries.  Round 4: the stages used to read W[k s], W[2 k s], W[3 k s] with s = n2 / (2 Ns) from the natural-order
// table — lane-dependent strides of 512 / 128 / 32 bytes for Ns = 4 / 16 / 64, i.e. 4- to 8-way bank conflicts on 9 of the ~50 LDS
// instructions of a row transform (PMC: SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.52 for k_x_inverse, 0.45 for k_x_forward; the
// data accesses are conflict-free by the swizzle above).  Consecutive k in consecutive 16-byte slots is conflict-free.
template <bool INV, int TW = 1>
__device__ __forceinline__ void xf_team_fft(double2 *__restrict__ row, int n2, int tid, bool active, const double2 *__restrict__ W,
                                            const double2 *__restrict__ Wst)
{
    const int T = n2 >> 2;
    int Ns = 1;
    int woff = 0;
    const bool r3 = (n2 % 3) == 0;
    if (r3) {
        // n2 = 3 * 2^m (rows of 96, 192, 384, 768 cells — 768 x 768 x 256 is one of the reference's three CI grids): one radix-3 stage
        // first (Ns = 1, no twiddles): n2 / 3 = 4 T / 3 butterflies, thread tid takes butterfly tid and, if tid < T / 3, tid + T
        const int M = n2 / 3;
        const double S3 = 0.8660254037844386;      // sin(pi / 3)
        double2 o[2][3];
        bool has[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int j = tid + h * T;
            has[h] = active && j < M;
            o[h][0] = o[h][1] = o[h][2] = make_double2(0.0, 0.0);
            if (has[h]) {
                const double2 v0 = row[XF_P(j)], v1 = row[XF_P(j + M)], v2 = row[XF_P(j + 2 * M)];
                const double2 t1 = make_double2(v1.x + v2.x, v1.y + v2.y);
                const double2 t2 = make_double2(v0.x - 0.5 * t1.x, v0.y - 0.5 * t1.y);
                const double2 t3 = make_double2(S3 * (v1.x - v2.x), S3 * (v1.y - v2.y));
                const double2 lo = make_double2(t2.x + t3.y, t2.y - t3.x), hi = make_double2(t2.x - t3.y, t2.y + t3.x);      // t2 -/+ i t3
                o[h][0] = make_double2(v0.x + t1.x, v0.y + t1.y);
                o[h][1] = INV ? hi : lo;
                o[h][2] = INV ? lo : hi;
            }
        }
        xf_sync<TW>();
#pragma unroll
        for (int h = 0; h < 2; ++h)
            if (has[h]) {
                const int j = tid + h * T;
                row[XF_P(3 * j)] = o[h][0]; row[XF_P(3 * j + 1)] = o[h][1]; row[XF_P(3 * j + 2)] = o[h][2];
            }
        xf_sync<TW>();
        Ns = 3;
    }
    for (; Ns * 4 <= n2; Ns <<= 2) {
        double2 o0 = make_double2(0.0, 0.0), o1 = o0, o2 = o0, o3 = o0;
        int j0 = 0;
        if (active) {
            const int k = r3 ? tid % Ns : tid & (Ns - 1);
            double2 v0 = row[XF_P(tid)], v1 = row[XF_P(tid + T)], v2 = row[XF_P(tid + 2 * T)], v3 = row[XF_P(tid + 3 * T)];
            if (Ns > 1) {
                double2 w1 = Wst[woff + k], w2 = Wst[woff + Ns + k], w3 = Wst[woff + 2 * Ns + k];
                if (INV) { w1.y = -w1.y; w2.y = -w2.y; w3.y = -w3.y; }
                v1 = xf_cmul(v1, w1); v2 = xf_cmul(v2, w2); v3 = xf_cmul(v3, w3);
            }
            const double2 a0 = make_double2(v0.x + v2.x, v0.y + v2.y), a1 = make_double2(v0.x - v2.x, v0.y - v2.y);
            const double2 a2 = make_double2(v1.x + v3.x, v1.y + v3.y), d = make_double2(v1.x - v3.x, v1.y - v3.y);
            const double2 a3 = INV ? make_double2(-d.y, d.x) : make_double2(d.y, -d.x);        // (v1 - v3) times +i / -i
            o0 = make_double2(a0.x + a2.x, a0.y + a2.y);
            o1 = make_double2(a1.x + a3.x, a1.y + a3.y);
            o2 = make_double2(a0.x - a2.x, a0.y - a2.y);
            o3 = make_double2(a1.x - a3.x, a1.y - a3.y);
            j0 = ((tid - k) << 2) + k;
        }
        xf_sync<TW>();
        if (active) { row[XF_P(j0)] = o0; row[XF_P(j0 + Ns)] = o1; row[XF_P(j0 + 2 * Ns)] = o2; row[XF_P(j0 + 3 * Ns)] = o3; }
        xf_sync<TW>();
        if (Ns > 1) woff += 3 * Ns;
    }
    if (Ns < n2) {      // n2 = 2 Ns: two radix-2 butterflies per thread, outputs land on their own inputs
        if (active) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int b = tid + h * T;
                double2 w = W[2 * b];
                if (INV) w.y = -w.y;
                const double2 u0 = row[XF_P(b)], u1 = xf_cmul(row[XF_P(b + Ns)], w);
                row[XF_P(b)] = make_double2(u0.x + u1.x, u0.y + u1.y);
                row[XF_P(b + Ns)] = make_double2(u0.x - u1.x, u0.y - u1.y);
            }
        }
        xf_sync<TW>();
    }
}

// Fill the stage-major twiddle table (n2 entries at most) from the natural-order one: called once per workgroup, by all threads.
__device__ __forceinline__ void xf_stage_twiddles(double2 *__restrict__ Wst, const double2 *__restrict__ Wg, int n2, int t0, int nthreads)
{
    int woff = 0;
    for (int Ns = ((n2 % 3) == 0) ? 3 : 1; Ns * 4 <= n2; Ns <<= 2) {
        if (Ns == 1) continue;
        const int s = n2 / (2 * Ns);
        for (int t = t0; t < 3 * Ns; t += nthreads) {
            const int m = t / Ns, k = t - m * Ns;
            Wst[woff + t] = Wg[(m + 1) * k * s];
        }
        woff += 3 * Ns;
    }
}
#define XF_WST_SLOTS(n2) (n2)
// entries of the natural-order table exp(-2 pi i t / Nx) a workgroup keeps in LDS: the split reads W[p], p <= n2 / 2; only the radix-2 stage of
// transforms whose length is not 3^a 4^b reads further (W[2 b], b < n2 / 2).  Round 5: rows of 512 cells (n2 = 256 = 4^4) keep 132 entries
// instead of 384 — 39.5 KB of LDS per workgroup instead of 43.4, i.e. FOUR workgroups per CU instead of three for kernels that spend most of
// their wave time parked on the load -> barrier -> stages -> store chain of a level (counter table)
__host__ __device__ inline int xf_w_entries(int n2)
{
    int m = n2;
    if (m % 3 == 0) m /= 3;
    while (m % 4 == 0) m /= 4;
    return (m == 1) ? (n2 / 2 + 4) : (3 * n2 / 2);      // m == 2: a radix-2 stage closes the transform
}

// Z = transform of the packed row z[n] = x[2n] + i x[2n+1]  ->  X[0 .. n2] of the real row, in place (row has n2 + 1 slots).
// Pairs (p, n2 - p) are independent: no barrier between a pair's reads and writes.
__device__ __forceinline__ void xf_split_forward(double2 *__restrict__ row, int n2, int tid, const double2 *__restrict__ W)
{
    const int T = n2 >> 2;
    for (int p = tid; p <= (n2 >> 1); p += T) {
        if (p == 0) {
            const double2 z = row[0];
            row[0] = make_double2(z.x + z.y, 0.0);
            row[XF_P(n2)] = make_double2(z.x - z.y, 0.0);
        } else {
            const int m = n2 - p;
            const double2 A = row[XF_P(p)], B = row[XF_P(m)];
            const double2 E = make_double2(0.5 * (A.x + B.x), 0.5 * (A.y - B.y));
            const double2 D = make_double2(0.5 * (A.x - B.x), 0.5 * (A.y + B.y));
            const double2 WD = xf_cmul(W[p], D);
            row[XF_P(p)] = make_double2(E.x + WD.y, E.y - WD.x);      // E - i W D
            row[XF_P(m)] = make_double2(E.x - WD.y, -(E.y + WD.x));   // conj(E + i W D)
        }
    }
}
// inverse of the above up to the factor 2 of the unnormalised transform: X[0 .. n2] -> Z'[0 .. n2-1]
__device__ __forceinline__ void xf_split_inverse(double2 *__restrict__ row, int n2, int tid, const double2 *__restrict__ W)
{
    const int T = n2 >> 2;
    for (int p = tid; p <= (n2 >> 1); p += T) {
        if (p == 0) {
            const double a = row[0].x, b = row[XF_P(n2)].x;
            row[0] = make_double2(a + b, a - b);
        } else {
            const int m = n2 - p;
            const double2 P = row[XF_P(p)], Q = row[XF_P(m)];
            const double2 S = make_double2(P.x + Q.x, P.y - Q.y);
            const double2 Dd = make_double2(P.x - Q.x, P.y + Q.y);
            double2 w = W[p];
            w.y = -w.y;
            const double2 V = xf_cmul(w, Dd);
            row[XF_P(p)] = make_double2(S.x - V.y, S.y + V.x);        // S + i V
            row[XF_P(m)] = make_double2(S.x + V.y, -(S.y - V.x));     // conj(S - i V)
        }
    }
}

// Where element (level k, wavenumber kx, row) of the transposed half spectrum lives: one array hatT[(k NXH + kx) Ny + row] on a single
// GPU (nkx = nxp = NXH, blk = 0); on a y-slab rank W blocks of nkx wavenumbers each, block d being the message for / from rank d
// (bz_comm.hip: all-to-all of the blocks), with the zero padding of the last block (nxp = W nkx >= NXH) written by k_x_forward.
// Round 6, single GPU: kxs != 0 — kx-major, hatT[kx kxs + k Ny + row] with kxs = Nz Ny: a range of wavenumbers is contiguous, which is what
// lets the y transforms and the vertical solves run chunk by chunk out of the Infinity Cache (bz_poisson.hip: bzi_xf_middle).
struct XfLayout {
    int nkx, nxp;
    long long blk;       // elements (double2) per block
    int klo, khi;        // level range of this launch (the slab driver pipelines level chunks with the all-to-all messages)
    long long kxs;       // 0: level-major; else elements between consecutive wavenumbers (one block only)
};
__device__ __forceinline__ long long xf_addr(const XfLayout &L, int Ny, int k, int kx, int row)
{
    if (L.kxs) return (long long)kx * L.kxs + (long long)k * Ny + row;
    const int d = kx / L.nkx, kxl = kx - d * L.nkx;
    return (long long)d * L.blk + ((long long)k * L.nkx + kxl) * Ny + row;
}

// SRC 1: source term from the predictor momentum (arithmetic of k_poisson_source_rows); SRC 0: rows of the contiguous rhs buffer.
// grid (Ny / XF_RB, ceil(Nz / kchunk)), block XF_RB * Nx / 8, dynamic LDS (XF_RB * (n2 + 4) + 3 n2 / 2 + n2) * sizeof(double2).
// Thread (row r, tid) owns the cell pairs i = 2 (tid + q Nx / 8), i + 1, q = 0 .. 3 — the four packed complex elements its first
// butterfly reads; wlo carries rho_w of the lower faces from the previous level's upper faces (the block marches in z).
template <int SRC, int TW = 1>
__global__ __launch_bounds__(XF_RB * 64 * TW) void k_x_forward(DevGrid g, const double *__restrict__ rhs, const double *__restrict__ ru,
                                                          const double *__restrict__ rv, const double *__restrict__ rw, double dt,
                                                          double2 *__restrict__ hatT, XfLayout L, const double2 *__restrict__ Wg, int kchunk)
{
    extern __shared__ double2 xf_sm[];
    const int Nx = g.Nx, n2 = Nx >> 1, T = n2 >> 2, RS = XF_ROW_STRIDE(n2), NXH = n2 + 1;
    const int nthreads = blockDim.x;
    const int r = threadIdx.x / T, tid = threadIdx.x - r * T;
    double2 *__restrict__ W = xf_sm + XF_RB * RS;
    const int wn = xf_w_entries(n2);
    double2 *__restrict__ Wst = W + wn;
    for (int t = threadIdx.x; t < wn; t += nthreads) W[t] = Wg[t];
    xf_stage_twiddles(Wst, Wg, n2, threadIdx.x, nthreads);
    __syncthreads();                                       // the twiddle tables are loaded by all waves
    int bj, bk;
    xf_block(bj, bk);
    const int j0 = bj * XF_RB, j = j0 + r;
    const int kbeg = L.klo + bk * kchunk, kend = min(kbeg + kchunk, L.khi);
    double2 *__restrict__ row = xf_sm + r * RS;
    // y-slab: row Ny is the neighbour rank's first row, delivered into the halo by the caller's exchange
    const long long jp = (j + 1 < g.Ny || !g.wrap_y) ? (long long)g.Sx : (long long)g.Sx * (1 - g.Ny);
    double wlo[8];
    if (SRC) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const long long n = g.idx(2 * (tid + T * q), j, kbeg);
            const xf_pair2 w2 = *(const xf_pair2 *)(rw + n);
            wlo[2 * q] = w2.x; wlo[2 * q + 1] = w2.y;
        }
    }
    for (int k = kbeg; k < kend; ++k) {
        if (SRC) {
            const double Ax = g.Ax[k], Ay = g.Ay[k], Az = g.Az, Vi = g.Vinv_c[k], dz = g.dzc[k];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int i = 2 * (tid + T * q);
                const long long n = g.idx(i, j, k);
                // the cell pair (i, i + 1) of every array as one element-aligned two-word request (round 4: the Float32 kernel moved 4 bytes per lane)
                const xf_pair2 up = *(const xf_pair2 *)(ru + n), vq = *(const xf_pair2 *)(rv + n), vpq = *(const xf_pair2 *)(rv + n + jp),
                               whq = *(const xf_pair2 *)(rw + n + g.Sxy);
                const double u0 = up.x, u1 = up.y, u2 = ru[n + ((i + 2 < Nx) ? 2 : 2 - Nx)];
                const double v0 = vq.x, v1 = vq.y, vp0 = vpq.x, vp1 = vpq.y;
                const double wh0 = whq.x, wh1 = whq.y;
                double a = Ax * u1 - Ax * u0, b = Ay * vp0 - Ay * v0, c = Az * wh0 - Az * wlo[2 * q];
                const double x0 = dz * (Vi * (a + b + c)) / dt;
                a = Ax * u2 - Ax * u1; b = Ay * vp1 - Ay * v1; c = Az * wh1 - Az * wlo[2 * q + 1];
                const double x1 = dz * (Vi * (a + b + c)) / dt;
                wlo[2 * q] = wh0; wlo[2 * q + 1] = wh1;
                row[XF_P(tid + T * q)] = make_double2(x0, x1);
            }
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int i = 2 * (tid + T * q);
                const long long m = (long long)i + (long long)Nx * ((long long)j + (long long)g.Ny * k);
                row[XF_P(tid + T * q)] = *(const double2 *)(rhs + m);           // i even, Nx even: 16-byte aligned
            }
        }
        xf_sync<TW>();
        xf_team_fft<false, TW>(row, n2, tid, true, W, Wst);
        xf_split_forward(row, n2, tid, W);
        __syncthreads();                                   // rows change hands: the transposed store reads all of them
        for (int e = threadIdx.x; e < L.nxp * XF_RB; e += nthreads) {
            const int kx = e / XF_RB, rr = e - kx * XF_RB;
            hatT[xf_addr(L, g.Ny, k, kx, j0 + rr)] = (kx < NXH) ? xf_sm[rr * RS + XF_P(kx)] : make_double2(0.0, 0.0);
        }
        __syncthreads();
    }
}

// Transposed half spectrum -> rows of phi in the contiguous buffer phi_c (Nx * Ny * Nz).  Same grid, block and LDS as k_x_forward.
// (16 rows per workgroup — 256-byte segments per kx — measured the same 0.54 ms per launch at 512^3 as 8 rows: not the segment size.)
template <int TW = 1>
__global__ __launch_bounds__(XF_RB * 64 * TW) void k_x_inverse(DevGrid g, const double2 *__restrict__ hatT, XfLayout L, double *__restrict__ phi_c,
                                                          const double2 *__restrict__ Wg, int kchunk)
{
    extern __shared__ double2 xf_sm[];
    const int Nx = g.Nx, n2 = Nx >> 1, T = n2 >> 2, RS = XF_ROW_STRIDE(n2), NXH = n2 + 1;
    const int nthreads = blockDim.x;
    const int r = threadIdx.x / T, tid = threadIdx.x - r * T;
    double2 *__restrict__ W = xf_sm + XF_RB * RS;
    const int wn = xf_w_entries(n2);
    double2 *__restrict__ Wst = W + wn;
    for (int t = threadIdx.x; t < wn; t += nthreads) W[t] = Wg[t];
    xf_stage_twiddles(Wst, Wg, n2, threadIdx.x, nthreads);
    int bj, bk;
    xf_block(bj, bk);
    const int j0 = bj * XF_RB;
    const int kbeg = L.klo + bk * kchunk, kend = min(kbeg + kchunk, L.khi);
    double2 *__restrict__ row = xf_sm + r * RS;
    for (int k = kbeg; k < kend; ++k) {
        for (int e = threadIdx.x; e < NXH * XF_RB; e += nthreads) {
            const int kx = e / XF_RB, rr = e - kx * XF_RB;
            xf_sm[rr * RS + XF_P(kx)] = hatT[xf_addr(L, g.Ny, k, kx, j0 + rr)];
        }
        __syncthreads();                                   // rows loaded by all waves
        xf_split_inverse(row, n2, tid, W);
        xf_sync<TW>();
        xf_team_fft<true, TW>(row, n2, tid, true, W, Wst);
        {   // every team stores its own row: 16-byte elements, consecutive lanes
            double *dst = phi_c + (long long)Nx * ((long long)(j0 + r) + (long long)g.Ny * k);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int c = tid + T * q;
                *(double2 *)(dst + 2 * c) = row[XF_P(c)];
            }
        }
        __syncthreads();                                   // before the next level's loads overwrite the rows
    }
}
