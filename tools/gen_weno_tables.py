#!/usr/bin/env python
"""tools/gen_weno_tables.py — derive the finite-volume WENO tables of order 2r-1, r = 2 .. 5, in exact rational arithmetic and write them
as C tables for the oracle (oracle/weno_tables.h) and for the device (breeze.jl_amd/csrc/bz_weno_tables.h).

Conventions (the ones of oracle/breeze_oracle.c: weno5 and of csrc/bz_weno.h): the value is reconstructed at the face x_{i+1/2} of the
upwind cell i; stencil s = 0 .. r-1 holds the cells i-s .. i-s+r-1 (s = 0 is the most downwind one); the big stencil is
v[0 .. 2r-2] = cells i-(r-1) .. i+(r-1), so stencil s starts at v[r-1-s].
  C[s][j]   reconstruction coefficients of stencil s (Shu 1998, table 2.1: c_{rj} with r = s)
  D[s]      optimal (linear) weights
  B[s][j][l], l >= j: smoothness indicator beta_s = sum_j w_j (sum_{l>=j} B[s][j][l] w_l), w = the stencil's cells; the Jiang-Shu
            integral  sum_{m=1}^{r-1} int dx^(2m-1) (d^m p / dx^m)^2 dx  over the upwind cell, scaled to the integer tables in use:
            x1 (r = 2), x3 (r = 3: the table 10, -31, 11, 25, -19, 4 of the order-5 code), x240 (r = 4) and x5040 (r = 5), the
            Balsara & Shu (2000) tables that Oceananigans tabulates (recalled, not verified: parity unpinned)
  TAU[s]    WENO-Z global indicator tau = |sum_s TAU[s] beta_s|: (1,-1), (1,0,-1), (1,3,-3,-1), (1,2,-6,2,1)
Also the finite-volume centred interpolation coefficients of order 2, 4, 6, 8 (the advecting-flux interpolation Centered(order 2r-2)).
"""
import os
import sys
from fractions import Fraction as Fr

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def padd(a, b):
    n = max(len(a), len(b))
    return [(a[i] if i < len(a) else 0) + (b[i] if i < len(b) else 0) for i in range(n)]


def pmul(a, b):
    out = [Fr(0)] * (len(a) + len(b) - 1)
    for i, x in enumerate(a):
        for j, y in enumerate(b):
            out[i + j] += x * y
    return out


def pder(a):
    return [a[i] * i for i in range(1, len(a))] or [Fr(0)]


def pint(a, lo, hi):
    tot = Fr(0)
    for i, c in enumerate(a):
        tot += c * (Fr(hi) ** (i + 1) - Fr(lo) ** (i + 1)) / (i + 1)
    return tot


def peval(a, x):
    return sum(c * Fr(x) ** i for i, c in enumerate(a))


def cell_basis(r, s):
    """Polynomials p_j (degree r-1) with cell average 1 on cell j of the stencil (cells i-s .. i-s+r-1, dx = 1, cell i = [-1/2, 1/2])
    and 0 on the others: derivative of the primitive's Lagrange interpolant through the cell edges."""
    edges = [Fr(-s) - Fr(1, 2) + m for m in range(r + 1)]
    basis = []
    for j in range(r):
        # primitive values at the edges for unit average in cell j: 0 up to edge j, 1 from edge j+1 on
        vals = [Fr(0) if m <= j else Fr(1) for m in range(r + 1)]
        P = [Fr(0)]
        for m in range(r + 1):
            if vals[m] == 0:
                continue
            L = [Fr(1)]
            for q in range(r + 1):
                if q != m:
                    L = pmul(L, [-edges[q] / (edges[m] - edges[q]), Fr(1) / (edges[m] - edges[q])])
            P = padd(P, [vals[m] * c for c in L])
        basis.append(pder(P))
    return basis


def tables(r):
    scale = {2: 1, 3: 3, 4: 240, 5: 5040}[r]
    C, B = [], []
    for s in range(r):
        basis = cell_basis(r, s)
        C.append([peval(p, Fr(1, 2)) for p in basis])
        Q = [[Fr(0)] * r for _ in range(r)]
        for j in range(r):
            for l in range(r):
                dj, dl = basis[j], basis[l]
                tot = Fr(0)
                for m in range(1, r):
                    dj, dl = pder(dj), pder(dl)
                    tot += pint(pmul(dj, dl), Fr(-1, 2), Fr(1, 2))
                Q[j][l] = tot
        Bs = [[Fr(0)] * r for _ in range(r)]
        for j in range(r):
            for l in range(j, r):
                Bs[j][l] = scale * (Q[j][l] if j == l else 2 * Q[j][l])
        B.append(Bs)
    # optimal weights: sum_s D[s] C[s] embedded in the big stencil = the order 2r-1 reconstruction
    big = cell_basis(2 * r - 1, r - 1)
    cbig = [peval(p, Fr(1, 2)) for p in big]
    D = [None] * r
    # cell v[2r-2] (i + r - 1) belongs to stencil 0 only, then v[2r-3] to stencils 0, 1, ...
    for s in range(r):
        cell = 2 * r - 2 - s                      # index in the big stencil
        acc = cbig[cell]
        for t in range(s):
            jj = cell - (r - 1 - t)
            if 0 <= jj < r:
                acc -= D[t] * C[t][jj]
        D[s] = acc / C[s][cell - (r - 1 - s)]
    assert sum(D) == 1
    for cell in range(2 * r - 1):                 # full consistency
        tot = sum(D[s] * C[s][cell - (r - 1 - s)] for s in range(r) if 0 <= cell - (r - 1 - s) < r)
        assert tot == cbig[cell], (r, cell)
    return C, D, B


TAU = {2: (1, -1), 3: (1, 0, -1), 4: (1, 3, -3, -1), 5: (1, 2, -6, 2, 1)}


def difference_form(r):
    """The same tables expressed through the first differences d_i = w_{i+1} - w_i of a stencil (beta and p - w_centre do not change when
    a constant is added to the cells): beta_s = sum_i d_i (sum_{k>=i} BD[s][i][k] d_k), p_s = w[s] + sum_i CD[s][i] d_i, where w[s] is the
    upwind cell (local index s in stencil s).  What the Float32 build evaluates: on a 300 K field the expanded form loses every digit."""
    C, D, B = tables(r)
    BD, CD = [], []
    for s in range(r):
        # symmetric matrix of the quadratic form, then T^t Q T with w_j = w_0 + sum_{i<j} d_i
        Q = [[(B[s][min(j, l)][max(j, l)] / (1 if j == l else 2)) for l in range(r)] for j in range(r)]
        for j in range(r):                        # invariance under constants: every row of Q sums to zero
            assert sum(Q[j]) == 0, (r, s, j)
        M = [[sum(Q[j][l] for j in range(i + 1, r) for l in range(k + 1, r)) for k in range(r - 1)] for i in range(r - 1)]
        BD.append([[(M[i][k] if i == k else (2 * M[i][k] if k > i else Fr(0))) for k in range(r - 1)] for i in range(r - 1)])
        assert sum(C[s]) == 1
        cd = []
        for i in range(r - 1):                    # w_j - w_s = sum_{i=s}^{j-1} d_i (j > s),  -sum_{i=j}^{s-1} d_i (j < s)
            if i >= s:
                cd.append(sum(C[s][j] for j in range(i + 1, r)))
            else:
                cd.append(-sum(C[s][j] for j in range(0, i + 1)))
        CD.append(cd)
    return BD, CD


def centered(order):
    """Finite-volume centred reconstruction at the face between cells -1 and 0 from `order` cells: coefficient of the pair at distance d."""
    h = order // 2
    basis = cell_basis(order, h - 1)              # cells i-(h-1) .. i+h around the face x_{i+1/2}
    c = [peval(p, Fr(1, 2)) for p in basis]
    assert all(c[j] == c[order - 1 - j] for j in range(order))
    return [c[h - 1 - d] for d in range(h)]       # d = 0: the two adjacent cells


def fmt(x):
    x = Fr(x)
    return f"{x.numerator}.0" if x.denominator == 1 else f"{x.numerator}.0 / {x.denominator}.0"


def emit(prefix, qual):
    out = []
    for r in range(2, 6):
        C, D, B = tables(r)
        for s in range(r):
            for j in range(r):
                for l in range(j, r):
                    assert B[s][j][l].denominator == 1, ("non-integer smoothness coefficient", r, s, j, l, B[s][j][l])
        out.append(f"/* order {2 * r - 1} (r = {r}) */")
        out.append(f"{qual} double {prefix}C{r}[{r}][{r}] = {{" + ", ".join("{" + ", ".join(fmt(x) for x in row) + "}" for row in C) + "};")
        out.append(f"{qual} double {prefix}D{r}[{r}] = {{" + ", ".join(fmt(x) for x in D) + "};")
        out.append(f"{qual} double {prefix}B{r}[{r}][{r}][{r}] = {{" + ", ".join(
            "{" + ", ".join("{" + ", ".join(fmt(x) for x in row) + "}" for row in Bs) + "}" for Bs in B) + "};")
        out.append(f"{qual} double {prefix}T{r}[{r}] = {{" + ", ".join(fmt(x) for x in TAU[r]) + "};")
        if r >= 3:
            BD, CD = difference_form(r)
            out.append(f"{qual} double {prefix}BD{r}[{r}][{r - 1}][{r - 1}] = {{" + ", ".join(
                "{" + ", ".join("{" + ", ".join(fmt(x) for x in row) + "}" for row in Bs) + "}" for Bs in BD) + "};   /* difference form */")
            out.append(f"{qual} double {prefix}CD{r}[{r}][{r - 1}] = {{" + ", ".join("{" + ", ".join(fmt(x) for x in row) + "}" for row in CD) + "};")
    for order in (2, 4, 6, 8):
        c = centered(order)
        out.append(f"{qual} double {prefix}S{order}[{order // 2}] = {{" + ", ".join(fmt(x) for x in c) + f"}};   /* Centered(order {order}) */")
    return "\n".join(out) + "\n"


def main():
    head = "/* GENERATED by tools/gen_weno_tables.py (exact rational arithmetic) - do not edit */\n"
    with open(os.path.join(ROOT, "oracle", "weno_tables.h"), "w") as f:
        f.write(head + emit("OGW_", "static const"))
    with open(os.path.join(ROOT, "breeze.jl_amd", "csrc", "bz_weno_tables.h"), "w") as f:
        f.write(head + "#pragma once\n" + emit("BZW_", "__device__ static const"))
    if "--print" in sys.argv:
        sys.stdout.write(emit("W", "const"))


if __name__ == "__main__":
    main()
