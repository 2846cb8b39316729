// Coulomb J and exact-exchange K on the dense ERI, one streaming pass (HBM-bound,
// 0.25 flop/byte: deliberately NOT reshaped into a GEMM).
//
// Replaces
//   XCSolver::compute_coulomb's cublasDgemv (src/dft_solver.cu:550-555):
//       J[c] = sum_r eri[r*N2 + c] * dm[r]          (OP_N on the row-major buffer)
//   the driver's cp.einsum('ijkl,jl->ik', eri4d, dm) (dft.py:218):
//       K[i][k] = sum_j sum_l eri[(i*n+j)*N2 + k*n+l] * dm[j*n+l]
//
// Layout: ERI is (N2, N2) row-major, row r = (i,j), column c = (k,l).
// A workgroup owns one i, a j-range, and a run of whole k-segments of columns
// (KB*n <= 1024 columns, 4 per thread, consecutive threads on consecutive
// columns).  It streams its rows once; per loaded element it does one FMA into
// the J column partial and one into T[k][l] = sum_j eri * dm[j][l].  At the end
// T is segment-summed over l through LDS into K[i][k].  Partials are written
// to slabs and summed in fixed order by the k_*_reduce kernels (deterministic).
#pragma once
#include <hip/hip_runtime.h>

#include "device_util.hpp"

namespace qcdft {

constexpr int JK_COLS = 1024; // columns per workgroup (4 per thread)

// VEC (n even: every ERI row is 16-byte aligned): each thread owns column pairs
// (2t, 2t+1) + 512q and reads them with one 16-byte load; otherwise single columns t + 256q.
template <bool WANT_J, bool WANT_K, bool VEC>
__global__ __launch_bounds__(256) void k_jk_stream(int n, int KB, int jsplit, int i0, int ni,
                                                   const double *__restrict__ eri,
                                                   const double *__restrict__ dm,
                                                   double *__restrict__ Jpart,
                                                   double *__restrict__ Kpart)
{
    __shared__ double T[JK_COLS];
    const size_t N2 = (size_t)n * n;
    const int tid = threadIdx.x;
    // rows (i, j) with i in [i0, i0 + ni): `eri` points at row (i0, 0) (the whole matrix when i0 = 0, ni = n;
    // a rank's row block of the dense-ERI sharding of SURVEY 8(e) otherwise)
    const int il = blockIdx.y / jsplit, js = blockIdx.y - il * jsplit, i = i0 + il;
    const int jper = (n + jsplit - 1) / jsplit;
    const int jlo = js * jper, jhi = min(n, jlo + jper);
    const int klo = blockIdx.x * KB, khi = min(n, klo + KB);
    const int ncol = (khi - klo) * n;       // columns of this block (even when VEC)
    const size_t cbase = (size_t)klo * n;   // first column

    double ja[4] = {0, 0, 0, 0}, ka[4] = {0, 0, 0, 0};
    int col[4], lcol[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        col[q] = VEC ? 2 * tid + (q & 1) + 512 * (q >> 1) : tid + 256 * q;
        lcol[q] = (col[q] < ncol) ? (col[q] % n) : 0; // l index of the column
    }
    for (int j = jlo; j < jhi; ++j) {
        const size_t r = (size_t)i * n + j;
        const double *row = eri + ((size_t)il * n + j) * N2 + cbase;
        const double dr = WANT_J ? dm[r] : 0.0;
        const double *drow = dm + (size_t)j * n;
        double e[4] = {0, 0, 0, 0};
        if (VEC) {
#pragma unroll
            for (int h = 0; h < 2; ++h)
                if (col[2 * h] < ncol) { // ncol even: the pair is inside together
                    const double2 v = *reinterpret_cast<const double2 *>(row + col[2 * h]);
                    e[2 * h] = v.x;
                    e[2 * h + 1] = v.y;
                }
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (col[q] < ncol) e[q] = row[col[q]];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (WANT_J) ja[q] += e[q] * dr;
            if (WANT_K) ka[q] += e[q] * drow[lcol[q]];
        }
    }
    if (WANT_J) {
        double *jp = Jpart + (size_t)blockIdx.y * N2 + cbase;
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (col[q] < ncol) jp[col[q]] = ja[q];
    }
    if (WANT_K) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (col[q] < ncol) T[col[q]] = ka[q];
        __syncthreads();
        if (tid < khi - klo) {
            double s = 0.0;
            const double *t = &T[tid * n];
            for (int l = 0; l < n; ++l) s += t[l];
            Kpart[((size_t)js * ni + il) * n + klo + tid] = s;
        }
    }
}

// J alone from an ERI that is SYMMETRIC as an (N2, N2) matrix ((ij|kl) = (kl|ij), which every real ERI is): only its upper
// triangle is streamed -- half the bytes of the pass above, which is all there is to a Coulomb build.  Element e = eri[r][c],
// c >= r, contributes e dm[r] to J[c] (column partials in registers, as above) and, for c > r, e dm[c] to J[r] (a row sum: a
// wave reduction per row, added up over the workgroup's four waves at the end).  Workgroup (b, i): rows (i, 0..n-1), columns of
// k-segments [b KB, b KB + KB); blocks left of the diagonal only write zeros.  Slabs: i = 0..n-1 hold the column partials,
// n + b the row partials of column block b; k_sum_slabs8 adds them in a fixed order.  Opt-in (DFT_SetOption "eri_symmetric"):
// DFT_ComputeCoulomb's contract is eri^T . vec(dm) for ANY matrix (dft_solver.cu:550-555), which this kernel does not honour.
template <bool VEC>
__global__ __launch_bounds__(256) void k_j_sym(int n, int KB, const double *__restrict__ eri, const double *__restrict__ dm,
                                               double *__restrict__ Jpart)
{
    __shared__ double rp[4][JK_COLS];
    const size_t N2 = (size_t)n * n;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.x, i = blockIdx.y;
    const int klo = b * KB, khi = min(n, klo + KB);
    const int ncol = (khi - klo) * n;
    const size_t cbase = (size_t)klo * n, rmin = (size_t)i * n;
    double *colpart = Jpart + (size_t)i * N2 + cbase;
    double *rowpart = Jpart + (size_t)(n + b) * N2 + rmin;
    int col[4];
    size_t g[4];
    double dcol[4], ja[4] = {0, 0, 0, 0};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        col[q] = VEC ? 2 * tid + (q & 1) + 512 * (q >> 1) : tid + 256 * q;
        g[q] = cbase + col[q];
        dcol[q] = col[q] < ncol ? dm[g[q]] : 0.0;
    }
    if (cbase + ncol <= rmin) {   // wholly below the diagonal
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (col[q] < ncol) colpart[col[q]] = 0.0;
        for (int j = tid; j < n; j += 256) rowpart[j] = 0.0;
        return;
    }
    // four rows per turn: their loads are in flight together and their four row sums go through the reduction side by side
    for (int j0 = 0; j0 < n; j0 += 4) {
        double e[4][4], dr[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int j = j0 + u;
            const bool live = j < n;
            const size_t r = rmin + (live ? j : 0);
            const double *row = eri + r * N2 + cbase;
            dr[u] = live ? dm[r] : 0.0;
#pragma unroll
            for (int q = 0; q < 4; ++q) e[u][q] = 0.0;
            if (VEC) {
#pragma unroll
                for (int h = 0; h < 2; ++h)
                    if (live && col[2 * h] < ncol && g[2 * h + 1] >= r) {   // the pair reaches the diagonal or lies right of it
                        const double2 v = *reinterpret_cast<const double2 *>(row + col[2 * h]);
                        e[u][2 * h] = v.x;
                        e[u][2 * h + 1] = v.y;
                    }
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (live && col[q] < ncol && g[q] >= r) e[u][q] = row[col[q]];
            }
        }
        double racc[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const size_t r = rmin + j0 + u;
            racc[u] = 0.0;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (g[q] >= r) ja[q] = fma(e[u][q], dr[u], ja[q]);      // (rows past n carry zeros)
                if (g[q] > r) racc[u] = fma(e[u][q], dcol[q], racc[u]);
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) racc[u] = row16_sum(racc[u]);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const double tot = (__shfl(racc[u], 0, 64) + __shfl(racc[u], 16, 64)) + (__shfl(racc[u], 32, 64) + __shfl(racc[u], 48, 64));
            if (lane == 0 && j0 + u < n) rp[wave][j0 + u] = tot;
        }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q)
        if (col[q] < ncol) colpart[col[q]] = ja[q];
    __syncthreads();
    for (int j = tid; j < n; j += 256) rowpart[j] = (rp[0][j] + rp[1][j]) + (rp[2][j] + rp[3][j]);
}

// J from the EIGHTH of the ERI that is unique when (ij|kl) = (ji|kl) = (ij|lk) = (kl|ij) and dm = dm^T (option
// "eri_symmetric" = 2): rows r = (i >= j), columns c = (k >= l), c <= r in the packed pair order P(a, b) = a (a + 1) / 2 + b.
// An element e = eri[r][c] adds e w_c dm_c to J_r (w = 2 off the diagonal of its pair, 1 on it: the transposed pair is not
// read) and, for c != r, e w_r dm_r to J_c.  Workgroup (b, g): packed columns [CB b, CB b + CB), packed rows [RC g, RC g + RC)
// -- every live workgroup (rows not wholly before its columns) owns the same RC x CB elements, whatever i: a decomposition
// by (i, column block) left the work of a workgroup proportional to i^2 and the chip to its heaviest ones (72 against 40 us).
// Thread t owns packed columns CB b + t + 256 q -- consecutive threads on consecutive pairs, which are consecutive in
// memory along a k-run (l = 0..k) -- column partials in registers, row sums by wave reduction, sixteen loads in flight.
// Slabs of packed J: g = 0..nchunk-1 column partials, nchunk + b row partials.  172 MB instead of 1.35 GB at Benzene/def2-SVP.
constexpr int JS8_RC = 64;
template <int CPT>   // packed columns per thread: 4, 2 or 1
__global__ __launch_bounds__(256) void k_j_sym8(int n, const double *__restrict__ eri, const double *__restrict__ dm,
                                                double *__restrict__ Jpart)
{
    constexpr int CB = 256 * CPT, RPT = 16 / CPT;
    __shared__ double rp[4][JS8_RC];
    const size_t N2 = (size_t)n * n;
    const int NPK = n * (n + 1) / 2, nchunk = (NPK + JS8_RC - 1) / JS8_RC;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.x, g = blockIdx.y;
    const int pbase = b * CB, r0 = g * JS8_RC, r1 = min(NPK, r0 + JS8_RC);   // rows [r0, r1)
    double *colpart = Jpart + (size_t)g * NPK + pbase;
    double *rowpart = Jpart + (size_t)(nchunk + b) * NPK + r0;
    int pk[CPT];
    size_t off[CPT];
    double dw[CPT], ja[CPT];
#pragma unroll
    for (int q = 0; q < CPT; ++q) {
        ja[q] = 0.0;
        pk[q] = pbase + tid + 256 * q;
        int k = (int)((sqrt(8.0 * (double)pk[q] + 1.0) - 1.0) * 0.5);
        while (k * (k + 1) / 2 > pk[q]) --k;
        while ((k + 1) * (k + 2) / 2 <= pk[q]) ++k;
        const int l = pk[q] - k * (k + 1) / 2;
        const bool valid = pk[q] < NPK;
        off[q] = valid ? (size_t)k * n + l : 0;
        dw[q] = valid ? dm[(size_t)k * n + l] * (k != l ? 2.0 : 1.0) : 0.0;
        if (!valid) pk[q] = 0x7FFFFFFF;   // never <= a row's pair index
    }
    if (pbase >= r1) {   // every column of the block lies beyond every row of the chunk
#pragma unroll
        for (int q = 0; q < CPT; ++q)
            if (pbase + tid + 256 * q < NPK) colpart[tid + 256 * q] = 0.0;
        for (int j = tid; j < r1 - r0; j += 256) rowpart[j] = 0.0;
        return;
    }
    // (i, j) of the chunk's first row
    int ri = (int)((sqrt(8.0 * (double)r0 + 1.0) - 1.0) * 0.5);
    while (ri * (ri + 1) / 2 > r0) --ri;
    while ((ri + 1) * (ri + 2) / 2 <= r0) ++ri;
    int rj = r0 - ri * (ri + 1) / 2;
    // two sets of RPT rows in turn: the next set's loads are in flight while this set goes through the reductions
    double e0[RPT][CPT], d0[RPT], e1[RPT][CPT], d1[RPT];
#define QCDFT_J8_LOAD(E, D, PBASE)                                                                         \
    _Pragma("unroll") for (int u = 0; u < RPT; ++u) {                                                      \
        const int P = (PBASE) + u;                                                                         \
        const bool live = P < r1;                                                                          \
        const double *row = eri + ((size_t)ri * n + rj) * N2;                                              \
        D[u] = live ? dm[(size_t)ri * n + rj] * (ri != rj ? 2.0 : 1.0) : 0.0;                              \
        _Pragma("unroll") for (int q = 0; q < CPT; ++q) E[u][q] = (live && pk[q] <= P) ? row[off[q]] : 0.0; \
        if (live) { if (++rj > ri) { ++ri; rj = 0; } }                                                     \
    }
#define QCDFT_J8_USE(E, D, PBASE)                                                                          \
    {                                                                                                      \
        double racc[RPT];                                                                                  \
        _Pragma("unroll") for (int u = 0; u < RPT; ++u) {                                                  \
            const int P = (PBASE) + u;                                                                     \
            racc[u] = 0.0;                                                                                 \
            _Pragma("unroll") for (int q = 0; q < CPT; ++q) {                                              \
                if (pk[q] < P) ja[q] = fma(E[u][q], D[u], ja[q]);                                          \
                racc[u] = fma(E[u][q], dw[q], racc[u]);                                                    \
            }                                                                                              \
        }                                                                                                  \
        _Pragma("unroll") for (int u = 0; u < RPT; ++u) racc[u] = row16_sum(racc[u]);                      \
        _Pragma("unroll") for (int u = 0; u < RPT; ++u) {                                                  \
            const double tot = (__shfl(racc[u], 0, 64) + __shfl(racc[u], 16, 64)) + (__shfl(racc[u], 32, 64) + __shfl(racc[u], 48, 64)); \
            if (lane == 0 && (PBASE) + u < r1) rp[wave][(PBASE) + u - r0] = tot;                           \
        }                                                                                                  \
    }
    QCDFT_J8_LOAD(e0, d0, r0)
    for (int P0 = r0;;) {
        if (P0 + RPT < r1) { QCDFT_J8_LOAD(e1, d1, P0 + RPT) }
        QCDFT_J8_USE(e0, d0, P0)
        P0 += RPT;
        if (P0 >= r1) break;
        if (P0 + RPT < r1) { QCDFT_J8_LOAD(e0, d0, P0 + RPT) }
        QCDFT_J8_USE(e1, d1, P0)
        P0 += RPT;
        if (P0 >= r1) break;
    }
#undef QCDFT_J8_LOAD
#undef QCDFT_J8_USE
#pragma unroll
    for (int q = 0; q < CPT; ++q)
        if (pbase + tid + 256 * q < NPK) colpart[tid + 256 * q] = ja[q];
    __syncthreads();
    for (int j = tid; j < r1 - r0; j += 256) rowpart[j] = (rp[0][j] + rp[1][j]) + (rp[2][j] + rp[3][j]);
}

// The packed J of k_j_sym8 from its slabs (fixed order, as k_sum_slabs8) straight into J[i][j] and J[j][i]
__global__ __launch_bounds__(256) void k_sum_slabs8_sym(int n, int nslab, const double *__restrict__ part, double *__restrict__ J)
{
    __shared__ double red[256];
    const size_t NPK = (size_t)n * (n + 1) / 2;
    const int el = threadIdx.x & 31, grp = threadIdx.x >> 5;
    const size_t e = (size_t)blockIdx.x * 32 + el;
    double s = 0.0;
    if (e < NPK) {
        int k = grp;
        for (; k + 8 * 15 < nslab; k += 8 * 16) {
            double v[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) v[q] = part[(size_t)(k + 8 * q) * NPK + e];
#pragma unroll
            for (int q = 0; q < 16; ++q) s += v[q];
        }
        for (; k < nslab; k += 8) s += part[(size_t)k * NPK + e];
    }
    red[threadIdx.x] = s;
    __syncthreads();
    if (grp == 0 && e < NPK) {
        const double *p = &red[el];
        const double tot = ((p[0] + p[32]) + (p[64] + p[96])) + ((p[128] + p[160]) + (p[192] + p[224]));
        int a = (int)((sqrt(8.0 * (double)e + 1.0) - 1.0) * 0.5);
        while ((size_t)a * (a + 1) / 2 > e) --a;
        while ((size_t)(a + 1) * (a + 2) / 2 <= e) ++a;
        const int c = (int)(e - (size_t)a * (a + 1) / 2);
        J[(size_t)a * n + c] = tot;
        J[(size_t)c * n + a] = tot;
    }
}

// out[e] = sum_s part[s*stride + e] in a fixed order: 32 elements x 8 slab groups per block.
__global__ __launch_bounds__(256) void k_sum_slabs8(size_t nelem, int nslab, size_t stride,
                                                    const double *__restrict__ part,
                                                    double *__restrict__ out)
{
    __shared__ double red[256];
    const int el = threadIdx.x & 31, grp = threadIdx.x >> 5;
    const size_t e = (size_t)blockIdx.x * 32 + el;
    double s = 0.0;
    if (e < nelem) {
        int k = grp; // 16 loads in flight per thread, summed in slab order (see k_reduce_slabs8)
        for (; k + 8 * 15 < nslab; k += 8 * 16) {
            double v[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) v[q] = part[(size_t)(k + 8 * q) * stride + e];
#pragma unroll
            for (int q = 0; q < 16; ++q) s += v[q];
        }
        for (; k < nslab; k += 8) s += part[(size_t)k * stride + e];
    }
    red[threadIdx.x] = s;
    __syncthreads();
    if (grp == 0 && e < nelem) {
        const double *p = &red[el];
        out[e] = ((p[0] + p[32]) + (p[64] + p[96])) + ((p[128] + p[160]) + (p[192] + p[224]));
    }
}

} // namespace qcdft
