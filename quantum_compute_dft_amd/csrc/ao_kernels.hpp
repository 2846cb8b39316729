// AO values and Cartesian AO gradients on the grid (row a1 of the hot path).
//
// Replaces the reference's calls into PySCF, dft.numint.eval_ao(mol, coords,
// deriv=0) at grid.py:38 and deriv=1 (planes [1:4]) at grid.py:30-31, whose
// outputs the driver uploads as d_ao (ngrid,nao) and d_ao_grad (3,ngrid,nao)
// (dft.py:136-142,155,172).  The reference has no GPU code for this step.
//
// Conventions (PySCF / libcint, real spherical GTOs):
//   phi = [sum_p c_p exp(-a_p r^2)] * S_lm(x,y,z),   r relative to the shell centre,
//   c_p carry primitive and contraction normalisation (host prepares them),
//   S_lm = real solid harmonics, order  p: x,y,z;  d: xy,yz,z2,xz,x2-y2;
//   f: m=-3..3  (y(3x2-y2), xyz, y(4z2-x2-y2), z(2z2-3x2-3y2), x(4z2-x2-y2), z(x2-y2), x(x2-3y2)).
//
// The kernel is HBM-WRITE bound (8*nao*(1 or 4) bytes per grid point against ~100 flops per
// primitive), and what matters is how the rows leave the chip.  A first version wrote
// 32-column chunks: with four planes in flight the partial 128-byte lines it left for the
// next chunk cost 2.6 TB/s (stores alone 200 us of 213 for Benzene/def2-SVP, compute 61 us).
// Now a workgroup owns 16 (or 8) grid points x a column block of <= 128 columns (the whole row for
// nao <= 126): every plane's tile is staged in LDS and leaves as complete rows, 16 lanes
// writing 256 contiguous bytes with 16-byte stores, all column groups of a row back to back.
// Benzene/def2-SVP, four planes: 122 us = 4.3 TB/s against 82.5 us for a plain fill of the same
// 524 MB (6.35 TB/s); the arithmetic alone is ~60 us and only partly hides behind the stores.
// Compute mapping: lane = (point = lane & 15, shell slot = lane >> 4), 16 shells per pass over
// the 4 waves; the host orders the shells of a block by (l, nprim) so the four shells a wave
// evaluates together mostly share one code path.  Shell data are per-lane loads (L1/L2 hits).
#pragma once
#include <hip/hip_runtime.h>
#include "device_util.hpp"

namespace qcdft {

constexpr int AO_MAX_L = 3;
constexpr double AO_EXP_CUT = 46.0;
constexpr int AO_CW = 126;   // max columns per block (4 planes x 16 x 127 doubles < 64 KB)

struct AoShell {
    double x, y, z;
    int l, nprim, off, ao;
};
struct AoChunk { // one column block: shells [shell_lo, shell_hi), columns [col_lo, col_lo + ncol)
    int shell_lo, shell_hi, col_lo, ncol;
};

template <bool GRAD>
__device__ __forceinline__ void ao_put(double *tile, int plane_sz, int idx, double S, double Sx,
                                       double Sy, double Sz, double R0, double R1, double dx,
                                       double dy, double dz)
{
    tile[idx] = R0 * S;
    if (GRAD) {
        const double t = R1 * S;
        tile[plane_sz + idx] = R0 * Sx + t * dx;
        tile[2 * plane_sz + idx] = R0 * Sy + t * dy;
        tile[3 * plane_sz + idx] = R0 * Sz + t * dz;
    }
}

// order[k], k in [shell_lo, shell_hi): shell indices of the block sorted by (l, nprim).
// ldt: LDS leading dimension (odd, >= widest block).  Dynamic LDS: (GRAD?4:1)*PT*ldt doubles of tile,
// then (TAB) a copy of the shell table: AoShell[nshell], exp[nprim], coef[nprim], order[nshell].
// PT = grid points per workgroup (16 or 8): 8 halves the LDS tile so more workgroups fit a CU.
//
// Persistent workgroups (tile = blockIdx.x, += gridDim.x) built so that the arithmetic of tile t+1
// runs while the stores of tile t drain: on gfx9 loads and stores share vmcnt and retire in
// order, so ANY global load in the arithmetic phase would wait for the previous tile's stores.
// Hence the shell table lives in LDS (TAB), the next tile's coordinates are fetched BEFORE the
// current tile's stores are issued, and the stores are unconditional and fixed in number (lanes
// past the block's columns / the grid's rows repeat a neighbour's identical store) so the compiler
// can wait for that prefetch with a counted vmcnt instead of vmcnt(0).  Ablations of the previous,
// non-persistent form, whose arithmetic read the shell table and coordinates from global memory
// (Benzene/def2-SVP, 4 planes): stores alone 114 us against 82.5 us for a plain fill, arithmetic
// alone 45 us, together 123 us; this form 102 us.
template <bool GRAD, bool VEC, int PT, bool TAB>
__global__ __launch_bounds__(256) void k_eval_ao(long ngrid, int nao, int nchunk, int ldt, int nshell,
                                                 int nprim_total,
                                                 const AoShell *__restrict__ sh_g,
                                                 const double *__restrict__ pexp_g,
                                                 const double *__restrict__ pcoef_g,
                                                 const AoChunk *__restrict__ chunks,
                                                 const int *__restrict__ order_g,
                                                 const double *__restrict__ coords,
                                                 double *__restrict__ ao,
                                                 double *__restrict__ grad)
{
    extern __shared__ double tile[];
    constexpr int SPW = 64 / PT, NSLOT = 256 / PT, TPR = 256 / PT; // shells per wave / per pass; store threads per row
    constexpr int JN = (AO_CW + 2 * TPR - 1) / (2 * TPR);           // store passes over a block's columns
    constexpr int NPL = GRAD ? 4 : 1;
    const int plane_sz = PT * ldt;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int pt = lane % PT, slot = wave * SPW + lane / PT; // PT points x NSLOT shell slots
    // store mapping: row = tid / TPR, columns 2*TPR*j + 2*(tid % TPR) + {0,1}
    const int srow = tid / TPR, sseg = tid % TPR;

    const AoShell *sh = sh_g;
    const double *pexp = pexp_g, *pcoef = pcoef_g;
    const int *order = order_g;
    if (TAB) {
        double *tab = tile + NPL * plane_sz;
        AoShell *sh_l = reinterpret_cast<AoShell *>(tab);
        double *ex_l = tab + 5 * nshell, *cf_l = ex_l + nprim_total; // sizeof(AoShell) = 40 = 5 doubles
        int *or_l = reinterpret_cast<int *>(cf_l + nprim_total);
        for (int i = tid; i < 5 * nshell; i += 256) tab[i] = reinterpret_cast<const double *>(sh_g)[i];
        for (int i = tid; i < nprim_total; i += 256) { ex_l[i] = pexp_g[i]; cf_l[i] = pcoef_g[i]; }
        for (int i = tid; i < nshell; i += 256) or_l[i] = order_g[i];
        sh = sh_l; pexp = ex_l; pcoef = cf_l; order = or_l;
        __syncthreads();
    }

    const long ntile = (ngrid + PT - 1) / PT;
    long tl = blockIdx.x;
    double px, py, pz;
    {
        const long g = min(tl * PT + pt, ngrid - 1);
        px = coords[3 * g]; py = coords[3 * g + 1]; pz = coords[3 * g + 2];
        asm volatile("" : "+v"(px), "+v"(py), "+v"(pz)); // settle the load here so no path enters the loop with it pending
    }
    // ONE loop over (tile, column block) items: a nested block loop makes the compiler drain vmcnt in
    // its preheader, i.e. wait for the previous tile's stores after all
    int ci = 0;
    while (tl < ntile) {
        const long g0 = tl * PT;
        const long grow = min(g0 + srow, ngrid - 1); // rows past the grid repeat the last row (same point, same values)
        double nx, ny, nz;
        {
            const AoChunk ch = chunks[ci];
            for (int k = ch.shell_lo + slot; k < ch.shell_hi; k += NSLOT) {
                const AoShell q = sh[order[k]];
                const double x = px - q.x, y = py - q.y, z = pz - q.z;
                const double r2 = x * x + y * y + z * z;
                double R0 = 0.0, R1 = 0.0;
                for (int p = 0; p < q.nprim; ++p) {
                    const double a = pexp[q.off + p];
                    const double t = a * r2;
                    // exp(-46) = 1e-20: a primitive that far down its tail adds nothing at fp64.  Tested per
                    // WAVE (16 consecutive grid points x 4 shells) so the branch is uniform: grid points come
                    // atom by atom and radial shell by radial shell, and most points are far from most centres.
                    if (__builtin_amdgcn_ballot_w64(t < AO_EXP_CUT) == 0) continue;
                    const double e = t < AO_EXP_CUT ? pcoef[q.off + p] * exp(-t) : 0.0;
                    R0 += e;
                    R1 -= 2.0 * a * e;
                }
                const int i0 = pt * ldt + (q.ao - ch.col_lo);
                if (q.l == 0) {
                    constexpr double c = 0.282094791773878143;
                    ao_put<GRAD>(tile, plane_sz, i0, c, 0, 0, 0, R0, R1, x, y, z);
                } else if (q.l == 1) {
                    constexpr double c = 0.488602511902919921;
                    ao_put<GRAD>(tile, plane_sz, i0 + 0, c * x, c, 0, 0, R0, R1, x, y, z);
                    ao_put<GRAD>(tile, plane_sz, i0 + 1, c * y, 0, c, 0, R0, R1, x, y, z);
                    ao_put<GRAD>(tile, plane_sz, i0 + 2, c * z, 0, 0, c, R0, R1, x, y, z);
                } else if (q.l == 2) {
                    constexpr double c = 1.092548430592079070, d = 0.315391565252520002,
                                     e = 0.546274215296039535;
                    ao_put<GRAD>(tile, plane_sz, i0 + 0, c * x * y, c * y, c * x, 0, R0, R1, x, y, z);
                    ao_put<GRAD>(tile, plane_sz, i0 + 1, c * y * z, 0, c * z, c * y, R0, R1, x, y, z);
                    ao_put<GRAD>(tile, plane_sz, i0 + 2, d * (2 * z * z - x * x - y * y), -2 * d * x,
                                 -2 * d * y, 4 * d * z, R0, R1, x, y, z);
                    ao_put<GRAD>(tile, plane_sz, i0 + 3, c * x * z, c * z, 0, c * x, R0, R1, x, y, z);
                    ao_put<GRAD>(tile, plane_sz, i0 + 4, e * (x * x - y * y), 2 * e * x, -2 * e * y, 0,
                                 R0, R1, x, y, z);
                } else {
                    constexpr double f3 = 0.590043589926643510, f2 = 2.890611442640554055,
                                     f1 = 0.457045799464465739, f0 = 0.373176332590115391,
                                     f2b = 1.445305721320277020;
                    const double xx = x * x, yy = y * y, zz = z * z;
                    ao_put<GRAD>(tile, plane_sz, i0 + 0, f3 * y * (3 * xx - yy), f3 * 6 * x * y,
                                 f3 * (3 * xx - 3 * yy), 0, R0, R1, x, y, z);
                    ao_put<GRAD>(tile, plane_sz, i0 + 1, f2 * x * y * z, f2 * y * z, f2 * x * z,
                                 f2 * x * y, R0, R1, x, y, z);
                    ao_put<GRAD>(tile, plane_sz, i0 + 2, f1 * y * (4 * zz - xx - yy), -2 * f1 * x * y,
                                 f1 * (4 * zz - xx - 3 * yy), 8 * f1 * y * z, R0, R1, x, y, z);
                    ao_put<GRAD>(tile, plane_sz, i0 + 3, f0 * z * (2 * zz - 3 * xx - 3 * yy),
                                 -6 * f0 * x * z, -6 * f0 * y * z, f0 * (6 * zz - 3 * xx - 3 * yy), R0,
                                 R1, x, y, z);
                    ao_put<GRAD>(tile, plane_sz, i0 + 4, f1 * x * (4 * zz - xx - yy),
                                 f1 * (4 * zz - 3 * xx - yy), -2 * f1 * x * y, 8 * f1 * x * z, R0, R1,
                                 x, y, z);
                    ao_put<GRAD>(tile, plane_sz, i0 + 5, f2b * z * (xx - yy), 2 * f2b * x * z,
                                 -2 * f2b * y * z, f2b * (xx - yy), R0, R1, x, y, z);
                    ao_put<GRAD>(tile, plane_sz, i0 + 6, f3 * x * (xx - 3 * yy), f3 * (3 * xx - 3 * yy),
                                 -6 * f3 * x * y, 0, R0, R1, x, y, z);
                }
            }

            lds_barrier();
            const bool last = ci + 1 == nchunk;
            const long tn = last ? tl + gridDim.x : tl;
            { // next item's coordinates: issued before this item's stores, consumed after them
                const long gn = min(tn * PT + pt, ngrid - 1);
                nx = coords[3 * gn]; ny = coords[3 * gn + 1]; nz = coords[3 * gn + 2];
            }
            {
                const size_t rbase = (size_t)grow * nao + ch.col_lo;
                const size_t plane = (size_t)ngrid * nao;
                const double *t0 = tile + srow * ldt;
                const int clast = (ch.ncol - 1) & ~1; // last pair start inside the block
#pragma unroll
                for (int j = 0; j < JN; ++j) {
                    const int c = min(2 * sseg + 2 * TPR * j, clast); // lanes past the block repeat its last pair
                    if (VEC) { // nao and col_lo even, 16-byte aligned outputs (host-checked): c + 1 < ncol always
                        *reinterpret_cast<double2 *>(ao + rbase + c) = make_double2(t0[c], t0[c + 1]);
                        if (GRAD) {
                            *reinterpret_cast<double2 *>(grad + rbase + c) = make_double2(t0[plane_sz + c], t0[plane_sz + c + 1]);
                            *reinterpret_cast<double2 *>(grad + plane + rbase + c) = make_double2(t0[2 * plane_sz + c], t0[2 * plane_sz + c + 1]);
                            *reinterpret_cast<double2 *>(grad + 2 * plane + rbase + c) = make_double2(t0[3 * plane_sz + c], t0[3 * plane_sz + c + 1]);
                        }
                    } else {
#pragma unroll
                        for (int d = 0; d < 2; ++d) {
                            const int cd = min(c + d, ch.ncol - 1);
                            ao[rbase + cd] = t0[cd];
                            if (GRAD) {
                                grad[rbase + cd] = t0[plane_sz + cd];
                                grad[plane + rbase + cd] = t0[2 * plane_sz + cd];
                                grad[2 * plane + rbase + cd] = t0[3 * plane_sz + cd];
                            }
                        }
                    }
                }
            }
            lds_barrier();
            ci = last ? 0 : ci + 1;
            tl = tn;
        }
        // take the prefetched coordinates HERE, outside every inner loop, where the wait is a counted
        // vmcnt (the stores stay in flight); left to the first use inside the shell loop the compiler
        // flushes vmcnt(0) in that loop's preheader
        asm volatile("" : "+v"(nx), "+v"(ny), "+v"(nz));
        px = nx; py = ny; pz = nz;
    }
}

template <int PT>
inline void launch_eval_ao_pt(hipStream_t st, long ngrid, int nao, int nchunk, int maxcol, bool vec, int num_cu,
                              int nshell, int nprim_total,
                              const AoShell *sh, const double *pexp, const double *pcoef,
                              const AoChunk *chunks, const int *order, const double *coords, double *ao,
                              double *grad)
{
    const int ldt = maxcol | 1;
    const size_t tile_b = sizeof(double) * (grad ? 4 : 1) * PT * ldt;
    const size_t tab_b = sizeof(AoShell) * nshell + 2 * sizeof(double) * nprim_total + sizeof(int) * nshell + 8;
    const bool tab = tile_b + tab_b <= 64 * 1024; // shell table in LDS while it fits the default dynamic-LDS limit
    const size_t lds = tile_b + (tab ? tab_b : 0);
    // persistent grid: as many workgroups as stay resident (LDS 160 KB, 32 waves per CU)
    const long ntile = (ngrid + PT - 1) / PT;
    const long per_cu = std::max<long>(1, std::min<long>(8, (160 * 1024) / (long)(lds + 512)));
    dim3 g((unsigned)std::min<long>(ntile, per_cu * num_cu));
#define QCDFT_AO(G, V, T) hipLaunchKernelGGL((k_eval_ao<G, V, PT, T>), g, dim3(256), lds, st, ngrid, nao, nchunk, ldt, nshell, nprim_total, sh, pexp, pcoef, chunks, order, coords, ao, grad)
    if (grad) {
        if (vec) { if (tab) QCDFT_AO(true, true, true); else QCDFT_AO(true, true, false); }
        else     { if (tab) QCDFT_AO(true, false, true); else QCDFT_AO(true, false, false); }
    } else {
        if (vec) { if (tab) QCDFT_AO(false, true, true); else QCDFT_AO(false, true, false); }
        else     { if (tab) QCDFT_AO(false, false, true); else QCDFT_AO(false, false, false); }
    }
#undef QCDFT_AO
}

inline void launch_eval_ao(hipStream_t st, long ngrid, int nao, int nchunk, int maxcol, bool vec, int pt, int num_cu,
                           int nshell, int nprim_total,
                           const AoShell *sh, const double *pexp, const double *pcoef,
                           const AoChunk *chunks, const int *order, const double *coords, double *ao,
                           double *grad)
{
    if (pt == 8) launch_eval_ao_pt<8>(st, ngrid, nao, nchunk, maxcol, vec, num_cu, nshell, nprim_total, sh, pexp, pcoef, chunks, order, coords, ao, grad);
    else         launch_eval_ao_pt<16>(st, ngrid, nao, nchunk, maxcol, vec, num_cu, nshell, nprim_total, sh, pexp, pcoef, chunks, order, coords, ao, grad);
}

} // namespace qcdft
