// The part of an SCF cycle between the Fock contributions and the next density, on the device, as launches queued behind
// the cycle's J / K / sweep: what dft.py:212-236 does on the host with numpy (Fock assembly, DIIS, eigh(F, S), density, energy
// terms) and what scf.py's host loop spent 0.44 of a 0.93 ms Benzene cycle on (QCDFT_SCF_PROFILE=1: transfers 0.06, Fock 0.02,
// DIIS 0.07, occupied-subspace rotation 0.26, density + energies 0.03).  J, K and Vxc are already in HBM when the cycle's
// kernels finish; nothing but eight scalars and a sequence word crosses PCIe afterwards.
//
//   k_tail_fock     F = H + J + (V + V^T)/2 - c_hf K/2 into the DIIS ring; F c and S c (c = occupied orbitals, dm = c c^T)
//   k_tail_err      e = F D S - S D F through the thin factors, into the ring; one new row of the DIIS Gram matrix
//   k_tail_mix      Pulay coefficients (a <= 9 x 9 system in one wave's registers, per workgroup), F_ext = sum c_k F_k
//   k_tail_gemm x2  F_ext U and A = U^T (F_ext U): a 16 x 16 tile per workgroup, the contraction split over four waves
//   k_tail_rot      (nao <= 128, nocc <= 32) ONE workgroup, operands in LDS: the rotation K (n_virt x n_occ) that
//                   block-diagonalises A, by the diagonally preconditioned fixed point of scf.OccupiedRotation; then the new
//                   S-orthonormal basis U' = U W,
//                       U_o' = (U_o + U_v K) L^-T V,   L L^T = 1 + K^T K,   V: Jacobi rotations that make L^-1 F_o L^-T diagonal
//                       U_v' = T + (T K) X K^T,        T = U_v - U_o K^T,   X = -L^-T (1 + L)^-1
//                   (the Cholesky form of the completion: no eigen-decomposition of K^T K, and
//                    (1 + K X^T K^T)(1 + K K^T)(1 + K X K^T) = 1 exactly)
//   k_tail_rot_big  (to nao 512, nocc 64) the same with operands in memory; its fixed-point steps are launches of their own
//                   (k_rb_qb, k_rb_r, k_rb_decide: state in memory, early-out when the iteration is over, status 3 = queue more)
//                   and so are its long contractions (k_rb_fop, k_tail_gemm)
//   k_tail_gemm     U' = U W
//   k_tail_density  dm' = c' c'^T, tr(dm' H), tr(dm' J)/2, -c_hf tr(dm' K)/4, |dm' - dm|; the last workgroup adds the
//                   row partials in a fixed order and publishes them (and the sweep's Exc) to host-mapped memory
//
// A cycle whose rotation is refused (first-order step above 0.5, no convergence, aufbau order in doubt) or that has
// no basis yet reports status 1 and leaves F_ext in the caller's buffer: the caller diagonalises it (LAPACK or
// hipSOLVER, as scf.OccupiedRotation._exact does), uploads the basis and calls DFT_ScfTailFinish.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <new>

#include "../../include/dft_solver.h"
#include "device_util.hpp"

using namespace qcdft;

namespace {

constexpr int TL_MAXN = 128;   // basis functions of the LDS-resident rotation kernel (k_tail_rot)
constexpr int TL_MAXO = 32;    // occupied orbitals of it (half a wave per Jacobi pair, LDS copies of the small matrices)
constexpr int TL_BIGN = 512;   // ... of the memory-resident one (k_tail_rot_big) and of every other kernel here
constexpr int TL_BIGO = 64;
constexpr int TL_SPACE = 8;    // DIIS ring
constexpr int TL_ROT_T = 512;  // threads of the rotation kernels (eight waves: 256 registers each)
constexpr int TL_LD = TL_MAXO + 1;
constexpr int TL_BLD = TL_BIGO + 1;

struct TailArgs {
    int n, no, slot, nhist, have_coef, rotate, max_inner;
    int hist[TL_SPACE];
    double coef[TL_SPACE];
    double c_hf, tol, canon_tol;
};

struct RbState;

struct RotLds {   // offsets (doubles) into the rotation kernel's dynamic LDS; -1 = the matrix stays in memory
    int km, bm, qm, rm, smalls;
};

struct TailDev {
    int n = 0, no = 0;
    const double *H = nullptr, *S = nullptr;
    double *U = nullptr, *Fx = nullptr, *eig = nullptr;       // caller's
    double *blob = nullptr;                                   // everything below
    double *Fb, *Eb, *Gb, *FC, *SC, *gpart, *FU, *A, *Unew, *Km, *Rm, *Qm, *Bm, *epart, *rden, *smalls, *Kfix, *KXg, *Kt, *Kt2, *K2;
    bool big = false;           // k_tail_rot_big: operands in memory
    RbState *state = nullptr;   // the memory-resident fixed point's state (device)
    int steps_hint = 6;         // fixed-point steps queued per DFT_ScfTailStep on that path
    TailArgs last{};            // of the last step: DFT_ScfTailMore continues it
    int *status = nullptr;     // [0] status, [1] inner steps, [2] Jacobi sweeps, [3] ticket
    double *h_out = nullptr, *h_out_dev = nullptr;            // host-mapped: 8 doubles + sequence word
    unsigned long seq = 0;
    RotLds lo{-1, -1, -1, -1, 0}; // where the fixed point's matrices live
    unsigned rot_lds = 0;      // dynamic LDS bytes of k_tail_rot
    hipStream_t stream = nullptr;
    char err[256] = {0};
};

// value of `v` in lane `l` (wave-uniform l), through SGPRs: no LDS crossbar
__device__ __forceinline__ double bcast(double v, int l)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), l), hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
    return __hiloint2double(hi, lo);
}
// sum over the wave, every lane gets it: DPP row sums and four lane reads (a ds_bpermute butterfly is six LDS round trips)
__device__ __forceinline__ double wave_sum(double v)
{
    v = row16_sum(v);
    return (bcast(v, 0) + bcast(v, 16)) + (bcast(v, 32) + bcast(v, 48));
}
__device__ __forceinline__ double wave_max(double v)
{
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
    return v;
}
// F row i into ring slot `slot`; (F c)[i, :] and (S c)[i, :].  Thread (o, q): orbital o, quarter q of the j range.
__global__ __launch_bounds__(256) void k_tail_fock(TailArgs a, const double *__restrict__ H, const double *__restrict__ S,
                                                   const double *__restrict__ J, const double *__restrict__ Kx,
                                                   const double *__restrict__ V, const double *__restrict__ c,
                                                   double *__restrict__ Fb, double *__restrict__ FC, double *__restrict__ SC,
                                                   int *__restrict__ status)
{
    __shared__ double frow[TL_BIGN], srow[TL_BIGN], pf[8][TL_BIGO], ps[8][TL_BIGO];
    const int n = a.n, no = a.no, i = blockIdx.x, t = threadIdx.x;
    if (i == 0 && t == 0) { status[0] = 0; status[1] = 0; status[2] = 0; }   // read by the kernels behind this one only
    double *F = Fb + (size_t)a.slot * n * n;
    for (int j = t; j < n; j += 256) {
        double f = H[i * n + j] + J[i * n + j] + 0.5 * (V[i * n + j] + V[j * n + i]);   // dft.py:212,223
        if (Kx) f -= 0.5 * a.c_hf * Kx[i * n + j];                                     // dft.py:221
        F[i * n + j] = f;
        frow[j] = f;
        srow[j] = S[i * n + j];
    }
    __syncthreads();
    // eight j ranges of 32 orbitals, or four of 64
    const bool wide = no > 32;
    const int o = wide ? t & 63 : t & 31, q = wide ? t >> 6 : t >> 5, nq = wide ? 4 : 8;
    const int per = (n + nq - 1) / nq, j0 = q * per, j1 = min(n, j0 + per);
    double fc = 0.0, sc = 0.0;
    if (o < no)
        for (int jb = j0; jb < j1; jb += 16) {   // sixteen orbital coefficients in flight, then their products
            double cj[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) cj[u] = jb + u < j1 ? c[(jb + u) * no + o] : 0.0;
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int j = jb + u < j1 ? jb + u : j0;   // (cj = 0 past the range)
                fc = fma(frow[j], cj[u], fc);
                sc = fma(srow[j], cj[u], sc);
            }
        }
    pf[q][o] = fc;
    ps[q][o] = sc;
    __syncthreads();
    if (t < no) {
        double x = (pf[0][t] + pf[1][t]) + (pf[2][t] + pf[3][t]), y = (ps[0][t] + ps[1][t]) + (ps[2][t] + ps[3][t]);
        if (!wide) { x += (pf[4][t] + pf[5][t]) + (pf[6][t] + pf[7][t]); y += (ps[4][t] + ps[5][t]) + (ps[6][t] + ps[7][t]); }
        FC[i * no + t] = x;
        SC[i * no + t] = y;
    }
}

// e[i, j] = (F D S - S D F)[i, j] with D = c c^T: sum_o FC[i,o] SC[j,o] - SC[i,o] FC[j,o]   (scf.CDIIS: sdf^T - sdf);
// partial Gram row: gpart[i][s] = sum_j Eb[hist[s]][i, j] e[i, j]
__global__ __launch_bounds__(128) void k_tail_err(TailArgs a, const double *__restrict__ FC, const double *__restrict__ SC,
                                                  double *__restrict__ Eb, double *__restrict__ gpart)
{
    __shared__ double fci[TL_BIGO], sci[TL_BIGO], part[2][TL_SPACE];
    const int n = a.n, no = a.no, i = blockIdx.x, t = threadIdx.x;
    if (t < TL_BIGO) { fci[t] = t < no ? FC[i * no + t] : 0.0; sci[t] = t < no ? SC[i * no + t] : 0.0; }
    __syncthreads();
    double *E = Eb + (size_t)a.slot * n * n;
    double gp[TL_SPACE];
#pragma unroll
    for (int s = 0; s < TL_SPACE; ++s) gp[s] = 0.0;
    for (int j = t; j < n; j += 128) {
        double e = 0.0;
        double eh[TL_SPACE];                          // the ring's rows: their loads go out with the orbitals' (nothing here depends on e)
#pragma unroll
        for (int s = 0; s < TL_SPACE; ++s) {
            const int h = a.hist[s < a.nhist ? s : 0];
            eh[s] = (s < a.nhist && h != a.slot) ? Eb[(size_t)h * n * n + (size_t)i * n + j] : 0.0;
        }
#pragma unroll
        for (int h0 = 0; h0 < TL_BIGO; h0 += 32) {   // thirty-two orbitals' loads first: one memory round trip, not n_occ of them
            if (h0 < no) {
                double sj[32], fj[32];
#pragma unroll
                for (int o = 0; o < 32; ++o) {
                    sj[o] = h0 + o < no ? SC[j * no + h0 + o] : 0.0;
                    fj[o] = h0 + o < no ? FC[j * no + h0 + o] : 0.0;
                }
#pragma unroll
                for (int o = 0; o < 32; ++o) e += fci[h0 + o] * sj[o] - sci[h0 + o] * fj[o];
            }
        }
        E[i * n + j] = e;
#pragma unroll
        for (int s = 0; s < TL_SPACE; ++s) {
            const bool self = s < a.nhist && a.hist[s] == a.slot;
            gp[s] = fma(self ? e : eh[s], e, gp[s]);
        }
    }
#pragma unroll
    for (int s = 0; s < TL_SPACE; ++s) {
        const double p = wave_sum(gp[s]);
        if ((t & 63) == 0) part[t >> 6][s] = p;
    }
    __syncthreads();
    if (t < a.nhist) gpart[i * TL_SPACE + t] = part[0][t] + part[1][t];
}

// Pulay coefficients (every workgroup solves the same <= 9 x 9 system, one row per lane, in registers) and row i of
// F_ext = sum_k c_k F_k.  status[0] = 2 if the system is singular.
__global__ __launch_bounds__(128) void k_tail_mix(TailArgs a, const double *__restrict__ gpart, double *__restrict__ Gb,
                                                  const double *__restrict__ Fb, double *__restrict__ Fx, int *__restrict__ status)
{
    __shared__ double cf[TL_SPACE];
    __shared__ int bad;
    const int n = a.n, t = threadIdx.x, i = blockIdx.x, m = a.nhist, lane = t & 63;
    if (t == 0) bad = 0;
    if (t < 64) {
        // the new Gram row: a fixed summation tree (every workgroup gets the same bits)
        double gs[TL_SPACE];
#pragma unroll
        for (int s = 0; s < TL_SPACE; ++s) {
            double p = 0.0;
            if (s < m)
                for (int r = lane; r < n; r += 64) p += gpart[r * TL_SPACE + s];
            gs[s] = wave_sum(p);
        }
        if (i == 0 && lane == 0)   // the ring's Gram matrix: row and column of the new slot (nobody reads those entries from memory here)
            for (int s = 0; s < m; ++s) Gb[a.slot * TL_SPACE + a.hist[s]] = Gb[a.hist[s] * TL_SPACE + a.slot] = gs[s];
        if (a.have_coef) {
            if (lane < m) cf[lane] = a.coef[lane];
        } else if (m < 2) {
            if (lane == 0) cf[0] = 1.0;
        } else {
            // B = [[0, 1^T], [1, G]], B x = e_0 (scf.CDIIS.update): Gaussian elimination with partial pivoting; lane p holds
            // row p and never moves it -- the pivot of step k is the largest |B[p][k]| among the rows not yet used
            const int d = m + 1;
            double row[TL_SPACE + 2];
            // lane p >= 1 is history entry p - 1; entries of the new slot's row / column come from this cycle's sums
            const int hp = a.hist[lane >= 1 && lane < d ? lane - 1 : 0];
            double gmine = 0.0;                                   // gs[lane - 1]
#pragma unroll
            for (int s = 0; s < TL_SPACE; ++s) gmine = s == lane - 1 ? gs[s] : gmine;
#pragma unroll
            for (int qq = 0; qq <= TL_SPACE; ++qq) {
                double v = 0.0;
                if (lane < d && qq < d) {
                    if (lane == 0 && qq == 0) v = 0.0;
                    else if (lane == 0 || qq == 0) v = 1.0;
                    else {
                        const int hq = a.hist[qq - 1];
                        v = hp == a.slot ? gs[qq - 1] : hq == a.slot ? gmine : Gb[hp * TL_SPACE + hq];
                    }
                }
                row[qq] = v;
            }
            row[TL_SPACE + 1] = lane == 0 ? 1.0 : 0.0;   // right-hand side
            bool used = false, sing = false;
            int order = -1;
#pragma unroll
            for (int k = 0; k <= TL_SPACE; ++k) {
                if (k < d && !sing) {
                    const double cand = (!used && lane < d) ? fabs(row[k]) : -1.0;
                    double mx = cand;                       // the candidates sit in lanes 0..8: a DPP maximum over the first 16-lane row
                    mx = fmax(mx, dpp_mov_f64<0x128>(mx)); mx = fmax(mx, dpp_mov_f64<0x124>(mx));
                    mx = fmax(mx, dpp_mov_f64<0x122>(mx)); mx = fmax(mx, dpp_mov_f64<0x121>(mx));
                    mx = bcast(mx, 0);
                    if (!(mx > 0.0)) { sing = true; }
                    else {
                        const int piv = __builtin_amdgcn_readfirstlane(__ffsll((long long)__ballot(cand == mx)) - 1);
                        double pr[TL_SPACE + 2];
#pragma unroll
                        for (int qq = 0; qq <= TL_SPACE + 1; ++qq) pr[qq] = bcast(row[qq], piv);
                        if (lane == piv) { used = true; order = k; }
                        else if (!used && lane < d) {
                            const double f = row[k] / pr[k];
#pragma unroll
                            for (int qq = 0; qq <= TL_SPACE + 1; ++qq) row[qq] -= f * pr[qq];
                        }
                    }
                }
            }
            double x[TL_SPACE + 1];
#pragma unroll
            for (int k = 0; k <= TL_SPACE; ++k) x[k] = 0.0;
            if (!sing) {
#pragma unroll
                for (int k = TL_SPACE; k >= 0; --k) {
                    if (k < d) {
                        const int own = __builtin_amdgcn_readfirstlane(__ffsll((long long)__ballot(order == k)) - 1);
                        double sacc = row[TL_SPACE + 1];
#pragma unroll
                        for (int qq = 0; qq <= TL_SPACE; ++qq)
                            if (qq > k && qq < d) sacc -= row[qq] * x[qq];
                        x[k] = bcast(sacc / row[k], own);
                        if (!(fabs(x[k]) < 1e300)) sing = true;   // inf / NaN
                    }
                }
            }
            if (sing) { if (lane == 0) bad = 1; }
            else {
#pragma unroll
                for (int k = 1; k <= TL_SPACE; ++k)
                    if (lane == 0 && k < d) cf[k - 1] = x[k];
            }
        }
    }
    __syncthreads();
    if (bad) {
        if (i == 0 && t == 0) status[0] = 2;
        return;
    }
    for (int j = t; j < n; j += 128) {
        double fb[TL_SPACE];
#pragma unroll
        for (int s = 0; s < TL_SPACE; ++s) fb[s] = s < m ? Fb[(size_t)a.hist[s] * n * n + (size_t)i * n + j] : 0.0;
        double f = 0.0;
#pragma unroll
        for (int s = 0; s < TL_SPACE; ++s)
            if (s < m) f = fma(cf[s], fb[s], f);
        Fx[i * n + j] = f;
    }
}

// One 16 x 16 tile of C (M x N) = A B per workgroup on the fp64 matrix cores, the contraction split over four waves
// (thirty-two contraction steps per round: eight per wave, all operands loaded before the round's first MFMA):
// A(i, k) = a[i ars + k acs], B(k, j) = b[k brs + j bcs].
__global__ __launch_bounds__(256) void k_tail_gemm(int M, int N, int Kd, const double *__restrict__ a, int ars, int acs,
                                                   const double *__restrict__ b, int brs, int bcs, double *__restrict__ c, int ldc,
                                                   double alpha, int add_diag, const int *__restrict__ status)
{
    __shared__ double part[4][4][64];
    if (status[0] != 0) return;
    const int nt = (N + 15) >> 4, tile = blockIdx.x, i0 = (tile / nt) << 4, j0 = (tile % nt) << 4;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, li = lane & 15, kq = lane >> 4;
    const int ia = i0 + li, jb = j0 + li;
    const bool aok = ia < M, bok = jb < N;
    d4 acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
    for (int k0 = 0; k0 < Kd; k0 += 128) {
        double av[8], bv[8];
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const int k = k0 + 4 * (4 * s + wave) + kq;
            av[s] = aok && k < Kd ? a[(size_t)ia * ars + (size_t)k * acs] : 0.0;
            bv[s] = bok && k < Kd ? b[(size_t)k * brs + (size_t)jb * bcs] : 0.0;
        }
#pragma unroll
        for (int s = 0; s < 8; s += 2) {
            acc0 = mfma_f64(av[s], bv[s], acc0);
            acc1 = mfma_f64(av[s + 1], bv[s + 1], acc1);
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) part[wave][r][lane] = acc0[r] + acc1[r];
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = i0 + kq + 4 * r, col = j0 + li;
            if (row < M && col < N)
                c[(size_t)row * ldc + col] = alpha * ((part[0][r][lane] + part[1][r][lane]) + (part[2][r][lane] + part[3][r][lane])) +
                                             (add_diag && row == col ? 1.0 : 0.0);
        }
    }
}

// C (M x N, ldc) = alpha sum_k A(i, k) B(k, j) + beta D(i, j) by the whole workgroup on the fp64 matrix cores, one 16 x 16 tile
// per wave at a time; A(i, k) = a[i ars + k acs], B(k, j) = b[k brs + j bcs] -- transposes are strides.  LDS = true: every
// operand and the result live in the workgroup's LDS (ds_read / ds_write with 32-bit addresses; through generic pointers
// the same loads are flat instructions with 64-bit address arithmetic, measured 2-3x slower here); false: anywhere.
// Eight contraction steps' worth of operands are loaded ahead of their MFMAs, which alternate between two accumulators
// (a dependent fp64 MFMA waits out the 16 passes of the one before).  The caller puts a barrier between dependent products.
template <bool LDS> struct TailPtr;
template <> struct TailPtr<false> {
    typedef const double *ro;
    typedef double *rw;
};
template <> struct TailPtr<true> {
    typedef const __attribute__((address_space(3))) double *ro;
    typedef __attribute__((address_space(3))) double *rw;
};

template <bool LDS>
__device__ __forceinline__ void wg_gemm_t(int M, int N, int Kd, const double *a_, int ars, int acs, const double *b_, int brs, int bcs,
                                          double alpha, double beta, const double *d_, int ldd, double *c_, int ldc)
{
    typename TailPtr<LDS>::ro a = (typename TailPtr<LDS>::ro)a_, b = (typename TailPtr<LDS>::ro)b_, d = (typename TailPtr<LDS>::ro)d_;
    typename TailPtr<LDS>::rw c = (typename TailPtr<LDS>::rw)c_;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwave = blockDim.x >> 6;
    const int mt = (M + 15) >> 4, nt = (N + 15) >> 4;
    const int li = lane & 15, kq = lane >> 4;
    for (int tile = wave; tile < mt * nt; tile += nwave) {
        const int i0 = (tile / nt) << 4, j0 = (tile % nt) << 4;
        const int ia = i0 + li, jb = j0 + li;
        const bool aok = ia < M, bok = jb < N;
        const int aoff = (aok ? ia : 0) * ars, boff = (bok ? jb : 0) * bcs;
        d4 acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
        // two operand sets in turn: the next eight steps' loads are in flight while this set's MFMAs run
        double av0[8], bv0[8], av1[8], bv1[8];
#define QCDFT_TL_LOAD(AV, BV, K0)                                       \
    _Pragma("unroll") for (int s = 0; s < 8; ++s) {                     \
        const int k = (K0) + 4 * s + kq;                                \
        const bool kok = k < Kd;                                        \
        AV[s] = aok && kok ? a[aoff + k * acs] : 0.0;                   \
        BV[s] = bok && kok ? b[boff + k * brs] : 0.0;                   \
    }
#define QCDFT_TL_MMA(AV, BV)                                            \
    _Pragma("unroll") for (int s = 0; s < 8; s += 2) {                  \
        acc0 = mfma_f64(AV[s], BV[s], acc0);                            \
        acc1 = mfma_f64(AV[s + 1], BV[s + 1], acc1);                    \
    }
        QCDFT_TL_LOAD(av0, bv0, 0)
        for (int k0 = 0;;) {
            if (k0 + 32 < Kd) { QCDFT_TL_LOAD(av1, bv1, k0 + 32) }
            __builtin_amdgcn_sched_barrier(0);
            QCDFT_TL_MMA(av0, bv0)
            __builtin_amdgcn_sched_barrier(0);
            k0 += 32;
            if (k0 >= Kd) break;
            if (k0 + 32 < Kd) { QCDFT_TL_LOAD(av0, bv0, k0 + 32) }
            __builtin_amdgcn_sched_barrier(0);
            QCDFT_TL_MMA(av1, bv1)
            __builtin_amdgcn_sched_barrier(0);
            k0 += 32;
            if (k0 >= Kd) break;
        }
#undef QCDFT_TL_LOAD
#undef QCDFT_TL_MMA
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = i0 + kq + 4 * r, col = j0 + li;
            if (row < M && col < N) {
                double v = alpha * (acc0[r] + acc1[r]);
                if (beta != 0.0) v += beta * d[row * ldd + col];
                c[row * ldc + col] = v;
            }
        }
    }
}
__device__ __forceinline__ void wg_gemm(int M, int N, int Kd, const double *a, int ars, int acs, const double *b, int brs, int bcs,
                        double alpha, double beta, const double *d, int ldd, double *c, int ldc)
{
    wg_gemm_t<false>(M, N, Kd, a, ars, acs, b, brs, bcs, alpha, beta, d, ldd, c, ldc);
}
__device__ __forceinline__ void wg_gemm_lds(int M, int N, int Kd, const double *a, int ars, int acs, const double *b, int brs, int bcs,
                            double alpha, double beta, const double *d, int ldd, double *c, int ldc)
{
    wg_gemm_t<true>(M, N, Kd, a, ars, acs, b, brs, bcs, alpha, beta, d, ldd, c, ldc);
}

__device__ double block_max(double v, double *scratch)
{
    v = wave_max(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = v;
    __syncthreads();
    double m = scratch[0];
    for (int w = 1; w < (int)(blockDim.x >> 6); ++w) m = fmax(m, scratch[w]);
    return m;
}

// sum over the (at most 32) lanes of a wave that hold a column's rows; every lane gets the total
__device__ __forceinline__ double half_sum(double v)
{
    v = row16_sum(v);
    return bcast(v, 0) + bcast(v, 16);
}

// Cholesky P = L L^T (P = 1 + Pm symmetrised) by wave 0, left-looking, in LDS: lane i owns row i and subtracts
// sum_k<j L[i][k] L[j][k] from P[i][j] (row j is read by every lane at the same address: a broadcast); then L^-1 (wave 0) and
// (1 + L)^-1 (wave 1) by forward substitution, a column per lane.  Plain loops: the fully unrolled register form of this
// (32 x 32 / 2 steps with v_readlane broadcasts) ran 2.5x longer, on instruction fetch.  Rows and columns past n_occ: identity.
template <int MAXO>
__device__ __forceinline__ void chol_and_inverses(int no, const double *Pm, double (*Lm)[MAXO + 1], double (*Li)[MAXO + 1], double (*L1)[MAXO + 1])
{
    const int t = threadIdx.x;
    for (int e = t; e < MAXO * MAXO; e += blockDim.x) {
        const int i = e / MAXO, j = e - i * MAXO;
        Lm[i][j] = (i < no && j < no) ? 0.5 * (Pm[i * no + j] + Pm[j * no + i]) + (i == j ? 1.0 : 0.0) : (i == j ? 1.0 : 0.0);
        Li[i][j] = 0.0;
        L1[i][j] = 0.0;
    }
    __syncthreads();
    if (t < 64) {
        const int lane = t, row = lane < MAXO ? lane : MAXO - 1;
        for (int j = 0; j < no; ++j) {
            double s = Lm[row][j], s2 = 0.0;
            int k = 0;
            for (; k + 4 <= j; k += 4) {   // four products' loads in flight, two chains
                const double a0 = Lm[row][k], a1 = Lm[row][k + 1], a2 = Lm[row][k + 2], a3 = Lm[row][k + 3];
                const double b0 = Lm[j][k], b1 = Lm[j][k + 1], b2 = Lm[j][k + 2], b3 = Lm[j][k + 3];
                s = fma(-a0, b0, s); s2 = fma(-a1, b1, s2); s = fma(-a2, b2, s); s2 = fma(-a3, b3, s2);
            }
            for (; k < j; ++k) s = fma(-Lm[row][k], Lm[j][k], s);
            s += s2;
            const double djj = sqrt(bcast(s, j));
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // every lane has read column j's inputs
            if (lane == j) Lm[j][j] = djj;
            else if (lane > j && lane < no) Lm[lane][j] = s / djj;
            else if (lane < j) Lm[lane][j] = 0.0;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // ... and sees column j before the next one (one wave: LDS in order)
        }
    }
    __syncthreads();
    if (t < 128) {
        const int lane = t & 63, col = lane < MAXO ? lane : MAXO - 1;
        const double plus = t >= 64 ? 1.0 : 0.0;                // (1 + L) has the same strictly lower part
        double (*Out)[MAXO + 1] = t >= 64 ? L1 : Li;
        for (int i = 0; i < no; ++i) {
            double s = col == i ? 1.0 : 0.0, s2 = 0.0;                  // Out[k][col] = 0 above the diagonal
            int k = 0;
            for (; k + 4 <= i; k += 4) {
                const double a0 = Lm[i][k], a1 = Lm[i][k + 1], a2 = Lm[i][k + 2], a3 = Lm[i][k + 3];
                const double b0 = Out[k][col], b1 = Out[k + 1][col], b2 = Out[k + 2][col], b3 = Out[k + 3][col];
                s = fma(-a0, b0, s); s2 = fma(-a1, b1, s2); s = fma(-a2, b2, s); s2 = fma(-a3, b3, s2);
            }
            for (; k < i; ++k) s = fma(-Lm[i][k], Out[k][col], s);
            s += s2;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (lane < MAXO) Out[i][col] = col <= i ? s / (Lm[i][i] + plus) : 0.0;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        if (lane < MAXO)
            for (int i = no; i < MAXO; ++i) Out[i][col] = col == i ? 1.0 / (1.0 + plus) : 0.0;
    }
    __syncthreads();
}

// 1 / x and 1 / sqrt(x) from the hardware estimates plus one Newton step (the Jacobi rotations need c^2 + s^2 = 1 to
// rounding, not correctly rounded quotients; the library division and square root are ~25 instructions each)
__device__ __forceinline__ double fast_rcp(double x)
{
    const double r = __builtin_amdgcn_rcp(x);
    return fma(fma(-x, r, 1.0), r, r);
}
__device__ __forceinline__ double fast_rsqrt(double x)
{
    const double y = __builtin_amdgcn_rsq(x);
    return y * fma(-0.5 * x * y, y, 1.5);
}

__global__ __launch_bounds__(TL_ROT_T) void k_tail_rot(TailArgs a, RotLds lo, const double *__restrict__ A, const double *__restrict__ U,
                                                       double *Km_g, double *Rm_g, double *Qm_g, double *Bm_g, double *W,
                                                       double *eig, int *status, long long *stamps)
{
#define QCDFT_STAMP(k) do { if (threadIdx.x == 0) stamps[k] = (long long)wall_clock64(); } while (0)
    // dynamic LDS: [A (n x n) during the fixed point | the completion's small matrices afterwards][K][B][Q][R] as they fit
    extern __shared__ __attribute__((aligned(16))) double dyn[];
    __shared__ double dd[TL_MAXN], red[TL_ROT_T / 64];
    __shared__ int flag[TL_ROT_T / 64];
    const int n = a.n, no = a.no, nv = n - no, t = threadIdx.x, nk = nv * no;
    if (status[0] != 0) return;   // the DIIS system was singular: nothing to rotate
    QCDFT_STAMP(0);
    if (threadIdx.x == 0) stamps[11] = (long long)clock64();
    double *As = dyn;
    double *Km = lo.km >= 0 ? dyn + lo.km : Km_g, *Bm = lo.bm >= 0 ? dyn + lo.bm : Bm_g, *Qm = lo.qm >= 0 ? dyn + lo.qm : Qm_g,
           *Rm = lo.rm >= 0 ? dyn + lo.rm : Rm_g;
    {   // 16-byte loads, sixteen in flight per thread (A and the LDS region start on 16-byte boundaries)
        const int npair = (n * n) >> 1;
        const double2 *A2 = reinterpret_cast<const double2 *>(A);
        double2 *As2 = reinterpret_cast<double2 *>(As);
        for (int e0 = 0; e0 < npair; e0 += 16 * TL_ROT_T) {
            double2 v[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) v[k] = e0 + k * TL_ROT_T + t < npair ? A2[e0 + k * TL_ROT_T + t] : make_double2(0.0, 0.0);
#pragma unroll
            for (int k = 0; k < 16; ++k)
                if (e0 + k * TL_ROT_T + t < npair) As2[e0 + k * TL_ROT_T + t] = v[k];
        }
        if ((n & 1) && t == 0) As[n * n - 1] = A[n * n - 1];
    }
    __syncthreads();
    const double *Aoo = As, *Aov = As + no, *Avo = As + (size_t)no * n, *Avv = As + (size_t)no * n + no;
    if (t < n) dd[t] = As[(size_t)t * n + t];
    __syncthreads();
    // this thread's elements of K (e = t, t + 512, ...): their denominators stay in registers
    constexpr int EPT = (TL_MAXN * TL_MAXO + TL_ROT_T - 1) / TL_ROT_T;   // nv no < 128 * 32
    double rd[EPT];
    double kmax = 0.0;
#pragma unroll
    for (int k = 0; k < EPT; ++k) {
        const int e = t + k * TL_ROT_T;
        rd[k] = 0.0;
        if (e < nk) {
            const int v = e / no, o = e - v * no;
            rd[k] = 1.0 / (dd[no + v] - dd[o]);
            const double x = -Avo[(size_t)v * n + o] * rd[k];
            Km[e] = x;
            kmax = fmax(kmax, fabs(x));
            if (!(fabs(x) <= 0.5)) kmax = 1.0;   // NaN too
        }
    }
    kmax = block_max(kmax, red);
    if (!(kmax <= 0.5)) {
        if (t == 0) status[0] = 1;
        return;
    }
    QCDFT_STAMP(1);
    // fixed point K <- K - R / (a_v - a_o),  R = Avo + Avv K - K (Aoo + Aov K)    (scf.OccupiedRotation)
    double prev = INFINITY;
    bool ok = false;
    int steps = 0;
    const bool all_lds = lo.km >= 0 && lo.bm >= 0 && lo.qm >= 0 && lo.rm >= 0;   // Benzene-sized problems: everything the loop touches
    for (int it = 0; it < a.max_inner; ++it) {
        if (all_lds) {
            wg_gemm_lds(nv, no, nv, Avv, n, 1, Km, no, 1, 1.0, 1.0, Avo, n, Qm, no);
            wg_gemm_lds(no, no, nv, Aov, n, 1, Km, no, 1, 1.0, 1.0, Aoo, n, Bm, no);
        } else {
            wg_gemm(nv, no, nv, Avv, n, 1, Km, no, 1, 1.0, 1.0, Avo, n, Qm, no);
            wg_gemm(no, no, nv, Aov, n, 1, Km, no, 1, 1.0, 1.0, Aoo, n, Bm, no);
        }
        __syncthreads();
        if (it == 0) QCDFT_STAMP(2);
        if (all_lds) wg_gemm_lds(nv, no, no, Km, no, 1, Bm, no, 1, -1.0, 1.0, Qm, no, Rm, no);
        else         wg_gemm(nv, no, no, Km, no, 1, Bm, no, 1, -1.0, 1.0, Qm, no, Rm, no);
        __syncthreads();
        if (it == 0) QCDFT_STAMP(3);
        double r = 0.0, rv[EPT];
#pragma unroll
        for (int k = 0; k < EPT; ++k) {
            const int e = t + k * TL_ROT_T;
            rv[k] = e < nk ? Rm[e] : 0.0;
            const double x = fabs(rv[k]);
            r = fmax(r, x);
            if (!(x == x)) r = INFINITY;
        }
        // the tentative update goes into the spare copy while the maximum is being agreed on (one barrier pair for both)
#pragma unroll
        for (int k = 0; k < EPT; ++k) {
            const int e = t + k * TL_ROT_T;
            if (e < nk) Rm[e] = Km[e] - rv[k] * rd[k];
        }
        r = block_max(r, red);
        ++steps;
        if (r < a.tol) { ok = true; break; }
        if (!(r < 4.0 * prev)) break;   // diverging (or NaN)
        prev = fmin(prev, r);
        { double *tmp = Km; Km = Rm; Rm = tmp; }   // K <- K - R / (a_v - a_o); the old K's space takes the next residual
        if (it == 0) QCDFT_STAMP(4);
    }
    QCDFT_STAMP(5);
    if (t == 0) status[1] = steps;
    if (!ok) {
        if (t == 0) status[0] = 1;
        return;
    }
    // The completion's small matrices: A's space is free now (K, B, Q stay where they are).
    const int n2o = no * no;
    double *sp = dyn + lo.smalls;
    double (*Lm)[TL_LD] = (double (*)[TL_LD])sp, (*Li)[TL_LD] = (double (*)[TL_LD])(sp + TL_MAXO * TL_LD),
           (*L1)[TL_LD] = (double (*)[TL_LD])(sp + 2 * TL_MAXO * TL_LD), (*Wc)[TL_LD] = (double (*)[TL_LD])(sp + 3 * TL_MAXO * TL_LD),
           (*Vc)[TL_LD] = (double (*)[TL_LD])(sp + 4 * TL_MAXO * TL_LD);
    double *cm = sp + 5 * TL_MAXO * TL_LD;
    double *Fo = cm, *Pm = cm + n2o, *W1 = cm + 2 * n2o, *Fop = cm + 3 * n2o, *Vo = cm + 4 * n2o, *cd = cm + 5 * n2o, *Xm = cm + 6 * n2o,
           *Lig = cm + 7 * n2o, *L1g = cm + 8 * n2o, *MX = cm + 9 * n2o;
    // Fo = Y^T F Y in the U basis (Y = Uo + Uv K), M = K^T K (P = Y^T Y = 1 + M)
    if (all_lds) {
        wg_gemm_lds(no, no, nv, Km, 1, no, Qm, no, 1, 1.0, 1.0, Bm, no, Fo, no);
        wg_gemm_lds(no, no, nv, Km, 1, no, Km, no, 1, 1.0, 0.0, Km, 0, Pm, no);
    } else {
        wg_gemm(no, no, nv, Km, 1, no, Qm, no, 1, 1.0, 1.0, Bm, no, Fo, no);
        wg_gemm(no, no, nv, Km, 1, no, Km, no, 1, 1.0, 0.0, nullptr, 0, Pm, no);
    }
    __syncthreads();
    QCDFT_STAMP(6);
    chol_and_inverses<TL_MAXO>(no, Pm, Lm, Li, L1);
    QCDFT_STAMP(7);
    for (int e = t; e < n2o; e += TL_ROT_T) {
        const int i = e / no, j = e - i * no;
        Lig[e] = Li[i][j];
        L1g[e] = L1[i][j];
    }
    __syncthreads();
    // G = L^-1 Fo L^-T
    wg_gemm_lds(no, no, no, Lig, no, 1, Fo, no, 1, 1.0, 0.0, Fo, 0, W1, no);
    __syncthreads();
    wg_gemm_lds(no, no, no, W1, no, 1, Lig, 1, no, 1.0, 0.0, Fo, 0, Fop, no);
    __syncthreads();
    QCDFT_STAMP(8);
    // One-sided Jacobi on the columns of G - sigma (sigma above the spectrum: all eigenvalues of one sign, so orthogonal
    // columns of (G - sigma) V are eigenvectors of G itself).  Half a wave per column pair (lane = row), the pairs of a
    // round by the circle method; a pair whose columns are already orthogonal to a tenth of the tolerance is left alone.
    const int ne = (no + 1) & ~1;
    {
        double gs = 0.0;
        if (t < no) {
            double rs = 0.0;
            for (int j = 0; j < no; ++j) rs += fabs(0.5 * (Fop[t * no + j] + Fop[j * no + t]));
            gs = rs;
        }
        const double sigma = block_max(gs, red) + 1.0;   // Gershgorin bound on |lambda|, plus a margin
        // An occupied block that is diagonal to canon_tol (Hartree) already is left as it is: the fixed point's denominators
        // only cross the gap, where off-diagonal elements of that size do not count (the virtual block is never diagonalised
        // at all), and the orbital energies reported are Rayleigh quotients, exact to second order.
        double off = 0.0;
        for (int e = t; e < n2o; e += TL_ROT_T)
            if (e / no != e % no) off = fmax(off, fabs(0.5 * (Fop[e] + Fop[(e % no) * no + e / no])));
        const bool skip_sweeps = block_max(off, red) <= a.canon_tol;
        for (int e = t; e < TL_MAXO * TL_MAXO; e += TL_ROT_T) {
            const int j = e / TL_MAXO, i = e - j * TL_MAXO;   // column j, row i
            double g = 0.0;
            if (i < no && j < no) g = 0.5 * (Fop[i * no + j] + Fop[j * no + i]) - (i == j ? sigma : 0.0);
            Wc[j][i] = g;
            Vc[j][i] = i == j ? 1.0 : 0.0;
        }
        __syncthreads();
        const int wave = t >> 6, lane = t & 63, npair = ne >> 1, row = lane & 31;
        // Rotations converge quadratically: a sweep that met no pair with |cos| above canon_tol leaves all of them below
        // ~canon_tol^2, which ends the iteration (canon_tol 1e-3: an occupied block diagonal to ~1e-6 of its scale -- the
        // next cycle's denominators want no more, and the orbitals are exactly orthonormal whatever the rotations were).
        const double stop2 = a.canon_tol * a.canon_tol, skip2 = 1e-26;
        int sweeps = 0;
        for (int sweep = 0; sweep < 12 && !skip_sweeps; ++sweep) {
            int big = 0;
            for (int round = 0; round < ne - 1; ++round) {
                const int pr = 2 * wave + (lane >> 5);   // sixteen pairs at a time
                if (2 * wave < npair) {
                    const bool live = pr < npair && row < ne;
                    // circle method: position 0 is held by column ne - 1, the others rotate
                    const int p = !live ? 0 : pr == 0 ? ne - 1 : (round + pr) % (ne - 1);
                    const int q = !live ? 0 : pr == 0 ? round : (round - pr + (ne - 1)) % (ne - 1);
                    const double wp = live ? Wc[p][row] : 0.0, wq = live ? Wc[q][row] : 0.0;
                    const double vp = live ? Vc[p][row] : 0.0, vq = live ? Vc[q][row] : 0.0;
                    double al = row16_sum(wp * wp), be = row16_sum(wq * wq), ga = row16_sum(wp * wq);
                    al += __shfl_xor(al, 16, 64); be += __shfl_xor(be, 16, 64); ga += __shfl_xor(ga, 16, 64);   // the pair's 32 lanes
                    const double g2 = ga * ga, ab = al * be;
                    if (live && g2 > skip2 * ab) {
                        big |= g2 > stop2 * ab;
                        const double zeta = 0.5 * (be - al) * fast_rcp(ga);
                        const double az = fabs(zeta);
                        const double hyp = az < 1e150 ? (1.0 + zeta * zeta) * fast_rsqrt(1.0 + zeta * zeta) : az;   // sqrt(1 + zeta^2)
                        const double tt = (zeta >= 0.0 ? 1.0 : -1.0) * fast_rcp(az + hyp);
                        const double cs = fast_rsqrt(1.0 + tt * tt), sn = cs * tt;
                        Wc[p][row] = cs * wp - sn * wq;
                        Wc[q][row] = sn * wp + cs * wq;
                        Vc[p][row] = cs * vp - sn * vq;
                        Vc[q][row] = sn * vp + cs * vq;
                    }
                }
                __syncthreads();
            }
            ++sweeps;
            const int any = __ballot(big) != 0;
            if (lane == 0) flag[wave] = any;
            __syncthreads();
            int more = 0;
            for (int k = 0; k < TL_ROT_T / 64; ++k) more |= flag[k];
            __syncthreads();
            if (!more) break;
        }
        if (t == 0) status[2] = sweeps;
        // eigenvalues lambda_j = v_j . (G v_j) = v_j . w_j + sigma
        for (int j = wave; j < no; j += TL_ROT_T / 64) {
            const double s = half_sum(lane < ne ? Vc[j][lane] * Wc[j][lane] : 0.0);
            if (lane == 0) dd[j] = s + sigma;   // the occupied slots of dd now hold the new orbital energies
        }
        for (int e = t; e < n2o; e += TL_ROT_T) {
            const int i = e / no, j = e - i * no;
            Vo[e] = Vc[j][i];
        }
        __syncthreads();
    }
    QCDFT_STAMP(9);
    // aufbau order: highest new occupied level against the lowest virtual diagonal (a necessary test only; the
    // caller checks the converged state against a full diagonalisation, as scf.py does)
    {
        double eo = -INFINITY, dv = INFINITY;
        if (t < no) eo = dd[t];
        else if (t < n) dv = dd[t];
        const double eomax = block_max(eo, red), dvmin = -block_max(-dv, red);
        if (eomax > dvmin - 1e-3) {
            if (t == 0) status[0] = 1;
            return;
        }
    }
    if (eig && t < n) eig[t] = dd[t];
    // The new basis as ONE product U' = U W (the next launch, on the whole chip):
    //   U_o' = (U_o + U_v K) c,   c = L^-T V                      ->  W[:, :no] = [c; K c]
    //   U_v' = T (1 + K X K^T),   T = U_v - U_o K^T, X = -L^-T (1 + L)^-1  ->  W[:, no:] = [-(1 + M X) K^T; 1 + (K X) K^T]
    double *Kc = Qm, *KX = Rm;   // Q and R are done with
    wg_gemm_lds(no, no, no, Lig, 1, no, Vo, no, 1, 1.0, 0.0, Vo, 0, cd, no);           // c = L^-T V
    wg_gemm_lds(no, no, no, Lig, 1, no, L1g, no, 1, -1.0, 0.0, Vo, 0, Xm, no);         // X = -L^-T (1 + L)^-1
    __syncthreads();
    if (all_lds) {
        wg_gemm_lds(nv, no, no, Km, no, 1, cd, no, 1, 1.0, 0.0, cd, 0, Kc, no);
        wg_gemm_lds(nv, no, no, Km, no, 1, Xm, no, 1, 1.0, 0.0, cd, 0, KX, no);
    } else {
        wg_gemm(nv, no, no, Km, no, 1, cd, no, 1, 1.0, 0.0, nullptr, 0, Kc, no);
        wg_gemm(nv, no, no, Km, no, 1, Xm, no, 1, 1.0, 0.0, nullptr, 0, KX, no);
    }
    wg_gemm_lds(no, no, no, Pm, no, 1, Xm, no, 1, 1.0, 0.0, cd, 0, MX, no);            // M X
    for (int e = t; e < n2o; e += TL_ROT_T) W[(size_t)(e / no) * n + e % no] = cd[e];
    __syncthreads();
    for (int e = t; e < n2o; e += TL_ROT_T) MX[e] += e / no == e % no ? 1.0 : 0.0;     // 1 + M X
    for (int e = t; e < nk; e += TL_ROT_T) W[(size_t)(no + e / no) * n + e % no] = Kc[e];
    __syncthreads();
    wg_gemm(no, nv, no, MX, no, 1, Km, 1, no, -1.0, 0.0, nullptr, 0, W + no, n);                      // -(1 + M X) K^T
    wg_gemm(nv, nv, no, KX, no, 1, Km, 1, no, 1.0, 0.0, nullptr, 0, W + (size_t)no * n + no, n);      // (K X) K^T ...
    __syncthreads();
    if (t < nv) W[(size_t)(no + t) * n + no + t] += 1.0;                                              // ... + 1
    QCDFT_STAMP(10);
    if (threadIdx.x == 0) stamps[12] = (long long)clock64();
#undef QCDFT_STAMP
}

// ---- the fixed point of the memory-resident rotation as launches of its own ------------------------------------------------
// At Anthracene's sizes a fixed-point step inside ONE workgroup costs 83 us (246 functions) to 250 us (494): its 90-odd
// tiles queue up on eight waves.  As three small launches per step they spread over the chip (~20 us per step); the
// iteration's state (converged / failed, step count, which of the two K buffers is current, the last residual) lives in
// memory, every kernel of a step returns at once when the iteration is over, and the host queues a few more steps than the
// last cycle needed (status 3 asks for more).
struct RbState {
    int done, ok, steps, cur;
    double prev;
    unsigned long long rmax_bits;
};

// one 16 x 16 tile of A B on four waves (the contraction dealt to them), summed into wave 0's registers; see k_tail_gemm
__device__ __forceinline__ void tile_splitk(int i0, int j0, int M, int N, int Kd, const double *a, int ars, int acs, const double *b, int brs,
                                            int bcs, double (*part)[4][64], double (&out)[4])
{
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, li = lane & 15, kq = lane >> 4;
    const int ia = i0 + li, jb = j0 + li;
    const bool aok = ia < M, bok = jb < N;
    d4 acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
    for (int k0 = 0; k0 < Kd; k0 += 128) {
        double av[8], bv[8];
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const int k = k0 + 4 * (4 * s + wave) + kq;
            av[s] = aok && k < Kd ? a[(size_t)ia * ars + (size_t)k * acs] : 0.0;
            bv[s] = bok && k < Kd ? b[(size_t)k * brs + (size_t)jb * bcs] : 0.0;
        }
#pragma unroll
        for (int s = 0; s < 8; s += 2) {
            acc0 = mfma_f64(av[s], bv[s], acc0);
            acc1 = mfma_f64(av[s + 1], bv[s + 1], acc1);
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) part[wave][r][lane] = acc0[r] + acc1[r];
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 4; ++r) out[r] = (part[0][r][lane] + part[1][r][lane]) + (part[2][r][lane] + part[3][r][lane]);
}

// Qt = Aov + K^T Avv (tiles [0, ntq)) and Bt = Aoo + K^T Avo (the rest), K = the current buffer
__global__ __launch_bounds__(256) void k_rb_qb(const RbState *__restrict__ st, int n, int no, const double *__restrict__ A,
                                               const double *K0, const double *K1, double *__restrict__ Qt, double *__restrict__ Bt)
{
    __shared__ double part[4][4][64];
    if (st->done) return;
    const int nv = n - no, nto = (no + 15) >> 4, ntv = (nv + 15) >> 4, ntq = nto * ntv;
    const double *K = st->cur ? K1 : K0;
    const double *Aoo = A, *Aov = A + no, *Avo = A + (size_t)no * n, *Avv = A + (size_t)no * n + no;
    const int lane = threadIdx.x & 63, li = lane & 15, kq = lane >> 4;
    double v[4];
    if ((int)blockIdx.x < ntq) {
        const int i0 = ((int)blockIdx.x / ntv) << 4, j0 = ((int)blockIdx.x % ntv) << 4;
        tile_splitk(i0, j0, no, nv, nv, K, 1, no, Avv, n, 1, part, v);
        if (threadIdx.x < 64)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = i0 + kq + 4 * r, col = j0 + li;
                if (row < no && col < nv) Qt[(size_t)row * nv + col] = v[r] + Aov[(size_t)row * n + col];
            }
    } else {
        const int tb = (int)blockIdx.x - ntq, i0 = (tb / nto) << 4, j0 = (tb % nto) << 4;
        tile_splitk(i0, j0, no, no, nv, K, 1, no, Avo, n, 1, part, v);
        if (threadIdx.x < 64)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = i0 + kq + 4 * r, col = j0 + li;
                if (row < no && col < no) Bt[(size_t)row * no + col] = v[r] + Aoo[(size_t)row * n + col];
            }
    }
}

// Rt = Qt - Bt Kt; the residual's maximum into the state; the tentative update K - R / (a_v - a_o) into the OTHER K buffers
__global__ __launch_bounds__(256) void k_rb_r(RbState *st, int n, int no, const double *__restrict__ Qt, const double *__restrict__ Bt,
                                              const double *__restrict__ rdt, double *K0, double *K1, double *Kt0, double *Kt1)
{
    __shared__ double part[4][4][64];
    if (st->done) return;
    const int nv = n - no, ntv = (nv + 15) >> 4;
    const int cur = st->cur;
    const double *Kt = cur ? Kt1 : Kt0;
    double *Kn = cur ? K0 : K1, *Ktn = cur ? Kt0 : Kt1;
    const int i0 = ((int)blockIdx.x / ntv) << 4, j0 = ((int)blockIdx.x % ntv) << 4;
    const int lane = threadIdx.x & 63, li = lane & 15, kq = lane >> 4;
    double v[4];
    tile_splitk(i0, j0, no, nv, no, Bt, no, 1, Kt, nv, 1, part, v);
    if (threadIdx.x < 64) {
        double rm = 0.0;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = i0 + kq + 4 * r, col = j0 + li;
            if (row < no && col < nv) {
                const size_t e = (size_t)row * nv + col;
                const double res = Qt[e] - v[r];
                const double x = fabs(res);
                rm = (x == x) ? fmax(rm, x) : INFINITY;
                const double kn = Kt[e] - res * rdt[e];
                Ktn[e] = kn;
                Kn[(size_t)col * no + row] = kn;
            }
        }
        rm = wave_max(rm);
        if (lane == 0) atomicMax(&st->rmax_bits, (unsigned long long)__double_as_longlong(rm));   // r >= 0: the bit patterns order like the values
    }
}

__global__ void k_rb_decide(RbState *st, double tol, int max_inner)
{
    if (st->done) return;
    const double r = __longlong_as_double((long long)st->rmax_bits);
    st->steps += 1;
    if (r < tol) { st->ok = 1; st->done = 1; return; }                 // the current K stands (Qt, Bt belong to it)
    if (!(r < 4.0 * st->prev) || st->steps >= max_inner) { st->done = 1; return; }   // diverging, NaN, or out of steps
    st->prev = fmin(st->prev, r);
    st->cur ^= 1;
    st->rmax_bits = 0ULL;
}

// When the iteration has converged: Fo^T = Bt + Q^T K (tiles [0, nto^2)) and M = K^T K (the rest) from the current K -- the
// completion's two long contractions (n_virt = hundreds), a tile per workgroup instead of nine tiles on one workgroup's waves
__global__ __launch_bounds__(256) void k_rb_fop(const RbState *__restrict__ st, int n, int no, const double *K0, const double *K1,
                                                const double *__restrict__ Qt, const double *__restrict__ Bt, double *__restrict__ Fo,
                                                double *__restrict__ Pm)
{
    __shared__ double part[4][4][64];
    if (!st->done || !st->ok) return;
    const int nv = n - no, nto = (no + 15) >> 4, nt2 = nto * nto;
    const double *K = st->cur ? K1 : K0;
    const bool isf = (int)blockIdx.x < nt2;
    const int tb = isf ? (int)blockIdx.x : (int)blockIdx.x - nt2, i0 = (tb / nto) << 4, j0 = (tb % nto) << 4;
    const int lane = threadIdx.x & 63, li = lane & 15, kq = lane >> 4;
    double v[4];
    if (isf) tile_splitk(i0, j0, no, no, nv, Qt, nv, 1, K, no, 1, part, v);
    else     tile_splitk(i0, j0, no, no, nv, K, 1, no, K, no, 1, part, v);
    if (threadIdx.x < 64)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = i0 + kq + 4 * r, col = j0 + li;
            if (row < no && col < no) {
                if (isf) Fo[row * no + col] = v[r] + Bt[row * no + col];
                else Pm[row * no + col] = v[r];
            }
        }
}

__global__ void k_tail_clear_more(int *status)
{
    if (status[0] == 3) status[0] = 0;
}

// The same rotation for sizes whose matrices do not fit the LDS (nao <= 512, nocc <= 64; Anthracene: 246 / 494 functions, 47
// occupied): one workgroup again -- a barrier across workgroups costs 4 us plus 0.3 us per workgroup on this chip
// (tools/gridsync_probe.hip, agent-scope fences either side), more than the phases it would separate -- with every operand in
// memory (L2-resident: A is 0.5-2 MB) behind generic pointers, the n_occ x n_occ triangular matrices alone in LDS.
__global__ __launch_bounds__(TL_ROT_T) void k_tail_rot_big(TailArgs a, const double *__restrict__ A, double *Km, double *K2, double *Kt,
                                                           double *Kt2, double *Qt, double *Rt, double *Bt, double *rdt, double *smalls,
                                                           double *W, double *eig, int *status, long long *stamps, double *Kfix, double *KXg,
                                                           int mode, RbState *st)
{
    // mode 0: everything here; 1: the start only (K0, its test, the iteration's state) -- the steps run as launches of
    // their own (k_rb_qb / k_rb_r / k_rb_decide); 2: the completion only, from the state those left
#define QCDFT_STAMP(k) do { if (threadIdx.x == 0) stamps[k] = (long long)wall_clock64(); } while (0)
    extern __shared__ double dyn[];   // L, L^-1, (1 + L)^-1; the Jacobi columns reuse the last two
    __shared__ double dd[TL_BIGN], red[TL_ROT_T / 64];
    __shared__ int flag[TL_ROT_T / 64];
    const int n = a.n, no = a.no, nv = n - no, t = threadIdx.x, nk = nv * no, n2o = no * no;
    if (status[0] != 0) return;
    QCDFT_STAMP(0);
    const double *Aoo = A, *Aov = A + no, *Avo = A + (size_t)no * n, *Avv = A + (size_t)no * n + no;
    for (int i = t; i < n; i += TL_ROT_T) dd[i] = A[(size_t)i * n + i];
    __syncthreads();
    bool ok = false;
    int steps = 0;
    if (mode != 2) {
    // K lives twice, as K[v][o] and as its transpose Kt[o][v]: with both, every large operand below is read along its rows
    // (16 lanes x 8 B from one 128-byte line; the strided alternative touches 16 lines per load and is bound by that, 4x slower)
    double kmax = 0.0;
    for (int e = t; e < nk; e += TL_ROT_T) {   // e = o nv + v: the transposed layout
        const int o = e / nv, v = e - o * nv;
        const double rd = 1.0 / (dd[no + v] - dd[o]);
        const double x = -Aov[(size_t)o * n + v] * rd;   // A is symmetric to rounding: Aov[o][v] for Avo[v][o]
        rdt[e] = rd;
        Kt[e] = x;
        Km[(size_t)v * no + o] = x;
        kmax = fmax(kmax, fabs(x));
        if (!(fabs(x) <= 0.5)) kmax = 1.0;
    }
    kmax = block_max(kmax, red);
    if (!(kmax <= 0.5)) {
        if (t == 0) { status[0] = 1; if (mode == 1) { st->done = 1; st->ok = 0; st->steps = 0; } }
        return;
    }
    QCDFT_STAMP(1);
    if (mode == 1) {
        if (t == 0) { st->done = 0; st->ok = 0; st->steps = 0; st->cur = 0; st->prev = INFINITY; st->rmax_bits = 0ULL; }
        return;
    }
    double prev = INFINITY;
    for (int it = 0; it < a.max_inner; ++it) {
        // Qt = (Avo + Avv K)^T = Aov + K^T Avv,  Bt = (Aoo + Aov K)^T = Aoo + K^T Avo   (A symmetric)
        wg_gemm(no, nv, nv, Km, 1, no, Avv, n, 1, 1.0, 1.0, Aov, n, Qt, nv);
        wg_gemm(no, no, nv, Km, 1, no, Avo, n, 1, 1.0, 1.0, Aoo, n, Bt, no);
        __syncthreads();
        if (it == 0) QCDFT_STAMP(2);
        // Rt = (Q - K B)^T = Qt - Bt Kt
        wg_gemm(no, nv, no, Bt, no, 1, Kt, nv, 1, -1.0, 1.0, Qt, nv, Rt, nv);
        __syncthreads();
        if (it == 0) QCDFT_STAMP(3);
        double r = 0.0;
        for (int e0 = 0; e0 < nk; e0 += 4 * TL_ROT_T) {   // four elements per thread in flight
            double rv[4], kv[4], dv[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int e = e0 + k * TL_ROT_T + t;
                rv[k] = e < nk ? Rt[e] : 0.0; kv[k] = e < nk ? Kt[e] : 0.0; dv[k] = e < nk ? rdt[e] : 0.0;
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int e = e0 + k * TL_ROT_T + t;
                const double x = fabs(rv[k]);
                r = fmax(r, x);
                if (!(x == x)) r = INFINITY;
                if (e < nk) {   // the tentative update, in both layouts
                    const double kn = kv[k] - rv[k] * dv[k];
                    Kt2[e] = kn;
                    K2[(size_t)(e % nv) * no + e / nv] = kn;
                }
            }
        }
        r = block_max(r, red);
        ++steps;
        if (r < a.tol) { ok = true; break; }
        if (!(r < 4.0 * prev)) break;
        prev = fmin(prev, r);
        { double *tmp = Km; Km = K2; K2 = tmp; tmp = Kt; Kt = Kt2; Kt2 = tmp; }
        if (it == 0) QCDFT_STAMP(4);
    }
    } else {   // mode 2: what the step launches left
        if (!st->done) {
            if (t == 0) status[0] = 3;   // more steps, please
            return;
        }
        ok = st->ok != 0;
        steps = st->steps;
        if (st->cur) { Km = K2; Kt = Kt2; }
    }
    QCDFT_STAMP(5);
    if (t == 0) status[1] = steps;
    if (!ok) {
        if (t == 0) status[0] = 1;
        return;
    }
    double (*Lm)[TL_BLD] = (double (*)[TL_BLD])dyn, (*Li)[TL_BLD] = (double (*)[TL_BLD])(dyn + TL_BIGO * TL_BLD),
           (*L1)[TL_BLD] = (double (*)[TL_BLD])(dyn + 2 * TL_BIGO * TL_BLD);
    double *Fo = smalls, *Pm = smalls + n2o, *W1 = smalls + 2 * n2o, *Fop = smalls + 3 * n2o, *Vo = smalls + 4 * n2o, *cd = smalls + 5 * n2o,
           *Xm = smalls + 6 * n2o, *Lig = smalls + 7 * n2o, *L1g = smalls + 8 * n2o, *MX = smalls + 9 * n2o;
    // Fo^T = Bt + Q^T K (the transpose does as well: only the symmetrised G = L^-1 Fo L^-T is used), M = K^T K
    if (mode != 2) {   // (mode 2: k_rb_fop has left both)
        wg_gemm(no, no, nv, Qt, nv, 1, Km, no, 1, 1.0, 1.0, Bt, no, Fo, no);
        wg_gemm(no, no, nv, Km, 1, no, Km, no, 1, 1.0, 0.0, nullptr, 0, Pm, no);
        __syncthreads();
    }
    QCDFT_STAMP(6);
    chol_and_inverses<TL_BIGO>(no, Pm, Lm, Li, L1);
    QCDFT_STAMP(7);
    for (int e = t; e < n2o; e += TL_ROT_T) {
        const int i = e / no, j = e - i * no;
        Lig[e] = Li[i][j];
        L1g[e] = L1[i][j];
    }
    __syncthreads();
    wg_gemm(no, no, no, Lig, no, 1, Fo, no, 1, 1.0, 0.0, nullptr, 0, W1, no);
    __syncthreads();
    wg_gemm(no, no, no, W1, no, 1, Lig, 1, no, 1.0, 0.0, nullptr, 0, Fop, no);
    __syncthreads();
    QCDFT_STAMP(8);
    {
        // Jacobi as in k_tail_rot, a whole wave per column pair (lane = row, up to 64 rows), eight pairs at a time
        double (*Wc)[TL_BLD] = Li, (*Vc)[TL_BLD] = L1;   // their contents are in memory now
        const int ne = (no + 1) & ~1, wave = t >> 6, lane = t & 63, npair = ne >> 1;
        double gs = 0.0, off = 0.0;
        if (t < no)
            for (int j = 0; j < no; ++j) gs += fabs(0.5 * (Fop[t * no + j] + Fop[j * no + t]));
        const double sigma = block_max(gs, red) + 1.0;
        for (int e = t; e < n2o; e += TL_ROT_T)
            if (e / no != e % no) off = fmax(off, fabs(0.5 * (Fop[e] + Fop[(e % no) * no + e / no])));
        const bool skip_sweeps = block_max(off, red) <= a.canon_tol;
        for (int e = t; e < TL_BIGO * TL_BIGO; e += TL_ROT_T) {
            const int j = e / TL_BIGO, i = e - j * TL_BIGO;
            double g = 0.0;
            if (i < no && j < no) g = 0.5 * (Fop[i * no + j] + Fop[j * no + i]) - (i == j ? sigma : 0.0);
            Wc[j][i] = g;
            Vc[j][i] = i == j ? 1.0 : 0.0;
        }
        __syncthreads();
        const double stop2 = a.canon_tol * a.canon_tol, skip2 = 1e-26;
        int sweeps = 0;
        for (int sweep = 0; sweep < 12 && !skip_sweeps; ++sweep) {
            int big = 0;
            for (int round = 0; round < ne - 1; ++round) {
                for (int pr = wave; pr < npair; pr += TL_ROT_T / 64) {
                    const bool live = lane < ne;
                    const int p = pr == 0 ? ne - 1 : (round + pr) % (ne - 1);
                    const int q = pr == 0 ? round : (round - pr + (ne - 1)) % (ne - 1);
                    const double wp = live ? Wc[p][lane] : 0.0, wq = live ? Wc[q][lane] : 0.0;
                    const double vp = live ? Vc[p][lane] : 0.0, vq = live ? Vc[q][lane] : 0.0;
                    const double al = wave_sum(wp * wp), be = wave_sum(wq * wq), ga = wave_sum(wp * wq);
                    const double g2 = ga * ga, ab = al * be;
                    if (live && g2 > skip2 * ab) {
                        big |= g2 > stop2 * ab;
                        const double zeta = 0.5 * (be - al) * fast_rcp(ga);
                        const double az = fabs(zeta);
                        const double hyp = az < 1e150 ? (1.0 + zeta * zeta) * fast_rsqrt(1.0 + zeta * zeta) : az;
                        const double tt = (zeta >= 0.0 ? 1.0 : -1.0) * fast_rcp(az + hyp);
                        const double cs = fast_rsqrt(1.0 + tt * tt), sn = cs * tt;
                        Wc[p][lane] = cs * wp - sn * wq;
                        Wc[q][lane] = sn * wp + cs * wq;
                        Vc[p][lane] = cs * vp - sn * vq;
                        Vc[q][lane] = sn * vp + cs * vq;
                    }
                }
                __syncthreads();
            }
            ++sweeps;
            const int any = __ballot(big) != 0;
            if (lane == 0) flag[wave] = any;
            __syncthreads();
            int more = 0;
            for (int k = 0; k < TL_ROT_T / 64; ++k) more |= flag[k];
            __syncthreads();
            if (!more) break;
        }
        if (t == 0) status[2] = sweeps;
        for (int j = wave; j < no; j += TL_ROT_T / 64) {
            const double sj = wave_sum(lane < ne ? Vc[j][lane] * Wc[j][lane] : 0.0);
            if (lane == 0) dd[j] = sj + sigma;
        }
        for (int e = t; e < n2o; e += TL_ROT_T) {
            const int i = e / no, j = e - i * no;
            Vo[e] = Vc[j][i];
        }
        __syncthreads();
    }
    QCDFT_STAMP(9);
    {
        double eo = -INFINITY, dv = INFINITY;
        for (int i = t; i < n; i += TL_ROT_T) {
            if (i < no) eo = fmax(eo, dd[i]);
            else dv = fmin(dv, dd[i]);
        }
        const double eomax = block_max(eo, red), dvmin = -block_max(-dv, red);
        if (eomax > dvmin - 1e-3) {
            if (t == 0) status[0] = 1;
            return;
        }
    }
    if (eig)
        for (int i = t; i < n; i += TL_ROT_T) eig[i] = dd[i];
    // W = [c, -(1 + M X) K^T; K c, 1 + (K X) K^T] (see k_tail_rot): the two left blocks and the small factors here, the two
    // right blocks (n_virt columns: hundreds of tiles) by the launches behind this kernel, which read K, K X and 1 + M X
    double *Kc = Qt;   // (n_virt x n_occ, row-major; Q is done with)
    wg_gemm(no, no, no, Lig, 1, no, Vo, no, 1, 1.0, 0.0, nullptr, 0, cd, no);
    wg_gemm(no, no, no, Lig, 1, no, L1g, no, 1, -1.0, 0.0, nullptr, 0, Xm, no);
    for (int e = t; e < nk; e += TL_ROT_T) Kfix[e] = Km[e];   // wherever the iteration left K
    __syncthreads();
    if (mode != 2) {   // (mode 2: K c straight into W and K X by tile launches behind this kernel)
        wg_gemm(nv, no, no, Kfix, no, 1, cd, no, 1, 1.0, 0.0, nullptr, 0, Kc, no);
        wg_gemm(nv, no, no, Kfix, no, 1, Xm, no, 1, 1.0, 0.0, nullptr, 0, KXg, no);
    }
    wg_gemm(no, no, no, Pm, no, 1, Xm, no, 1, 1.0, 0.0, nullptr, 0, MX, no);
    for (int e = t; e < n2o; e += TL_ROT_T) W[(size_t)(e / no) * n + e % no] = cd[e];
    __syncthreads();
    for (int e = t; e < n2o; e += TL_ROT_T) MX[e] += e / no == e % no ? 1.0 : 0.0;
    if (mode != 2)
        for (int e = t; e < nk; e += TL_ROT_T) W[(size_t)(no + e / no) * n + e % no] = Kc[e];
    QCDFT_STAMP(10);
#undef QCDFT_STAMP
}

// Row i of dm' = c' c'^T and of the energy traces; on success the new basis and orbitals replace the old ones.  The last
// workgroup to finish adds the row partials in a fixed order and publishes them.
__global__ __launch_bounds__(128) void k_tail_density(TailArgs a, int from_basis, unsigned long seq, const double *__restrict__ H,
                                                      const double *__restrict__ J, const double *__restrict__ Kx,
                                                      double *U, const double *Unew, double *dm, double *cocc,
                                                      double *epart, int *status, const double *exc, double *out)
{
    __shared__ double ci[TL_BIGO], part[2][4];
    __shared__ int last;
    const int n = a.n, no = a.no, i = blockIdx.x, t = threadIdx.x;
    const int st = from_basis ? 0 : status[0];
    const double sc = 1.4142135623730951;           // dm = 2 C_occ C_occ^T (dft.py:182): cocc = sqrt(2) C_occ
    const double *src = from_basis ? U : Unew;      // the caller's freshly diagonalised basis, or the rotated one
    if (st == 0) {
        if (t < TL_BIGO) ci[t] = t < no ? sc * src[(size_t)i * n + t] : 0.0;
        __syncthreads();
        double p1 = 0.0, p2 = 0.0, p3 = 0.0, p4 = 0.0;
        for (int j = t; j < n; j += 128) {
            const double dold = dm[i * n + j], h = H[i * n + j], jj = J[i * n + j], kk = Kx ? Kx[i * n + j] : 0.0;
            const double un = from_basis ? 0.0 : Unew[(size_t)i * n + j];
            double dn = 0.0;
#pragma unroll
            for (int h0 = 0; h0 < TL_BIGO; h0 += 32) {
                if (h0 < no) {
                    double cj[32];
#pragma unroll
                    for (int o = 0; o < 32; ++o) cj[o] = h0 + o < no ? sc * src[(size_t)j * n + h0 + o] : 0.0;
#pragma unroll
                    for (int o = 0; o < 32; ++o) dn = fma(ci[h0 + o], cj[o], dn);
                }
            }
            const double diff = dn - dold;
            p1 = fma(dn, h, p1); p2 = fma(dn, jj, p2); p3 = fma(dn, kk, p3); p4 = fma(diff, diff, p4);
            dm[i * n + j] = dn;
            if (!from_basis) U[(size_t)i * n + j] = un;   // nobody reads U in this mode: row i of the new basis replaces the old one
        }
        p1 = wave_sum(p1); p2 = wave_sum(p2); p3 = wave_sum(p3); p4 = wave_sum(p4);
        if ((t & 63) == 0) { part[t >> 6][0] = p1; part[t >> 6][1] = p2; part[t >> 6][2] = p3; part[t >> 6][3] = p4; }
        __syncthreads();
        if (t < 4) epart[4 * i + t] = part[0][t] + part[1][t];
        if (t < no) cocc[i * no + t] = ci[t];
    }
    __threadfence();
    __syncthreads();
    if (t == 0) last = atomicAdd(&status[3], 1) == n - 1;
    __syncthreads();
    if (!last) return;
    // every other workgroup has finished (its stores are visible: fence before the ticket); a fixed summation tree
    if (t < 64) {
        double s1 = 0.0, s2 = 0.0, s3 = 0.0, s4 = 0.0;
        if (st == 0)
            for (int r = t; r < n; r += 64) { s1 += epart[4 * r]; s2 += epart[4 * r + 1]; s3 += epart[4 * r + 2]; s4 += epart[4 * r + 3]; }
        s1 = wave_sum(s1); s2 = wave_sum(s2); s3 = wave_sum(s3); s4 = wave_sum(s4);
        if (t == 0) {
            out[0] = s1;                              // tr(dm' Hcore)
            out[1] = 0.5 * s2;                        // dft.py:233
            out[2] = Kx ? -0.25 * a.c_hf * s3 : 0.0;  // dft.py:234
            out[3] = sqrt(s4);                        // |dm' - dm|
            out[4] = (double)st;
            out[5] = (double)status[1];
            out[6] = (double)status[2];
            if (exc) out[7] = *exc;                   // Exc of the sweep queued before this step (DFT_ComputeXC*Async)
            status[3] = 0;
            __threadfence_system();
            ((volatile unsigned long *)out)[8] = seq;
        }
    }
}

// no rotation was asked for (no basis yet): the caller diagonalises F_ext
__global__ void k_tail_need_exact(int *status)
{
    if (status[0] == 0) status[0] = 1;
}

__global__ void k_tail_begin(int *status)
{
    status[0] = 0; status[1] = 0; status[2] = 0;
}

// The memory-resident rotation's completion: the two long contractions, the single-workgroup part, K c / K X / the wide blocks of W
void launch_big_finish(TailDev *c, const TailArgs &a)
{
    const int n = c->n, no = c->no, nv = n - no, nto = (no + 15) / 16, ntv = (nv + 15) / 16;
    hipStream_t st = c->stream;
    double *W = c->FU, *sm = c->smalls;
    const size_t n2o = (size_t)no * no;
    hipLaunchKernelGGL(k_rb_fop, dim3(2 * nto * nto), dim3(256), 0, st, c->state, n, no, c->Km, c->K2, c->Qm, c->Bm, sm, sm + n2o);
    hipLaunchKernelGGL(k_tail_rot_big, dim3(1), dim3(TL_ROT_T), c->rot_lds, st, a, c->A, c->Km, c->K2, c->Kt, c->Kt2, c->Qm, c->Rm, c->Bm,
                       c->rden, sm, W, c->eig, c->status, (long long *)(c->status + 8), c->Kfix, c->KXg, 2, c->state);
    hipLaunchKernelGGL(k_tail_gemm, dim3(ntv * nto), dim3(256), 0, st, nv, no, no, c->Kfix, no, 1, sm + 5 * n2o, no, 1,
                       W + (size_t)no * n, n, 1.0, 0, c->status);                                  // K c (W's lower left block)
    hipLaunchKernelGGL(k_tail_gemm, dim3(ntv * nto), dim3(256), 0, st, nv, no, no, c->Kfix, no, 1, sm + 6 * n2o, no, 1,
                       c->KXg, no, 1.0, 0, c->status);                                             // K X
    hipLaunchKernelGGL(k_tail_gemm, dim3(nto * ntv), dim3(256), 0, st, no, nv, no, sm + 9 * n2o, no, 1, c->Kfix, 1, no,
                       W + no, n, -1.0, 0, c->status);                                             // -(1 + M X) K^T
    hipLaunchKernelGGL(k_tail_gemm, dim3(ntv * ntv), dim3(256), 0, st, nv, nv, no, c->KXg, no, 1, c->Kfix, 1, no,
                       W + (size_t)no * n + no, n, 1.0, 1, c->status);                             // 1 + (K X) K^T
}

// `nsteps` fixed-point steps of the memory-resident rotation, three launches each (they return at once when the iteration is over)
void launch_big_steps(TailDev *c, const TailArgs &a, int nsteps)
{
    const int n = c->n, no = c->no, nv = n - no, nto = (no + 15) / 16, ntv = (nv + 15) / 16;
    for (int k = 0; k < nsteps; ++k) {
        hipLaunchKernelGGL(k_rb_qb, dim3(nto * ntv + nto * nto), dim3(256), 0, c->stream, c->state, n, no, c->A, c->Km, c->K2, c->Qm, c->Bm);
        hipLaunchKernelGGL(k_rb_r, dim3(nto * ntv), dim3(256), 0, c->stream, c->state, n, no, c->Qm, c->Bm, c->rden, c->Km, c->K2, c->Kt, c->Kt2);
        hipLaunchKernelGGL(k_rb_decide, dim3(1), dim3(1), 0, c->stream, c->state, a.tol, a.max_inner);
    }
}

void tail_error(TailDev *c, const char *what, hipError_t e)
{
    snprintf(c->err, sizeof c->err, "%s: %s", what, hipGetErrorString(e));
    fprintf(stderr, "libdft: %s\n", c->err);
}

} // namespace

extern "C" {

void *DFT_ScfTailOpen(int nao, int nocc, unsigned long long d_hcore, unsigned long long d_overlap, unsigned long long d_basis,
                      unsigned long long d_fock_out, unsigned long long d_mo_energy)
{
    if (nao < 2 || nao > TL_BIGN || nocc < 1 || nocc > TL_BIGO || nocc >= nao || !d_hcore || !d_overlap || !d_basis || !d_fock_out) return nullptr;
    TailDev *c = new (std::nothrow) TailDev();
    if (!c) return nullptr;
    const size_t n = (size_t)nao, no = (size_t)nocc, nv = n - no, n2 = n * n;
    c->n = nao; c->no = nocc;
    c->H = (const double *)d_hcore; c->S = (const double *)d_overlap;
    c->U = (double *)d_basis; c->Fx = (double *)d_fock_out; c->eig = (double *)d_mo_energy;
    c->big = nao > TL_MAXN || nocc > TL_MAXO;
    const size_t bigk = c->big ? nv * no : 0;
    const size_t sizes[] = {TL_SPACE * n2, TL_SPACE * n2, TL_SPACE * TL_SPACE, n * no, n * no, n * TL_SPACE, n2, n2, n2,
                            nv * no, nv * no, nv * no, no * no, 4 * n, bigk, c->big ? 10 * no * no : 0, bigk, bigk, bigk, bigk, bigk};
    size_t total = 0;
    for (size_t s : sizes) total += (s + 1) & ~(size_t)1;
    if (hipMalloc((void **)&c->blob, total * sizeof(double) + 512) != hipSuccess ||
        hipMemset(c->blob, 0, total * sizeof(double) + 512) != hipSuccess ||
        hipHostMalloc((void **)&c->h_out, 16 * sizeof(double), hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess ||
        hipHostGetDevicePointer((void **)&c->h_out_dev, c->h_out, 0) != hipSuccess) {
        (void)hipGetLastError();
        if (c->blob) (void)hipFree(c->blob);
        if (c->h_out) (void)hipHostFree(c->h_out);
        delete c;
        return nullptr;
    }
    double **slots[] = {&c->Fb, &c->Eb, &c->Gb, &c->FC, &c->SC, &c->gpart, &c->FU, &c->A, &c->Unew, &c->Km, &c->Rm, &c->Qm, &c->Bm, &c->epart,
                        &c->rden, &c->smalls, &c->Kfix, &c->KXg, &c->Kt, &c->Kt2, &c->K2};
    double *p = c->blob;
    for (size_t i = 0; i < sizeof(sizes) / sizeof(sizes[0]); ++i) {
        *slots[i] = p;
        p += (sizes[i] + 1) & ~(size_t)1;
    }
    c->status = (int *)p;
    c->state = (RbState *)(c->status + 40);   // [0..3] status words, [8..39] the rotation kernels' stamps, then the state
    memset(c->h_out, 0, 16 * sizeof(double));
    // k_tail_rot: A in LDS during the fixed point (the completion's small matrices reuse its space), then K, B, Q, R as 160 KB allow
    {
        const size_t smalls = 5 * (size_t)TL_MAXO * TL_LD + 10 * no * no, budget = (160 * 1024 - 2048) / sizeof(double);
        size_t off = (std::max(n2, smalls) + 1) & ~(size_t)1;
        const size_t want[4] = {nv * no, no * no, nv * no, nv * no};
        int *where[4] = {&c->lo.km, &c->lo.bm, &c->lo.qm, &c->lo.rm};
        for (int k = 0; k < 4; ++k) {
            const size_t sz = (want[k] + 1) & ~(size_t)1;
            if (off + sz <= budget) { *where[k] = (int)off; off += sz; }
        }
        c->rot_lds = (unsigned)(off * sizeof(double));
    }
    if (c->big) c->rot_lds = (unsigned)(3 * (size_t)TL_BIGO * TL_BLD * sizeof(double));
    if (hipFuncSetAttribute(c->big ? (const void *)k_tail_rot_big : (const void *)k_tail_rot, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)c->rot_lds) != hipSuccess) {
        (void)hipGetLastError();
        DFT_ScfTailClose(c);
        return nullptr;
    }
    return c;
}

void DFT_ScfTailClose(void *h)
{
    TailDev *c = (TailDev *)h;
    if (!c) return;
    (void)hipStreamSynchronize(c->stream);
    if (c->blob) (void)hipFree(c->blob);
    if (c->h_out) (void)hipHostFree(c->h_out);
    delete c;
}

int DFT_ScfTailSetStream(void *h, unsigned long long hip_stream)
{
    TailDev *c = (TailDev *)h;
    if (!c) return -1;
    (void)hipStreamSynchronize(c->stream);
    c->stream = (hipStream_t)hip_stream;
    return 0;
}

const char *DFT_ScfTailLastError(void *h)
{
    TailDev *c = (TailDev *)h;
    return c ? c->err : "null handle";
}

int DFT_ScfTailStep(void *h, int rotate, double c_hf, double tol, double canon_tol, int max_inner, int slot, int nhist, const int *hist,
                    const double *coef, unsigned long long d_J, unsigned long long d_K, unsigned long long d_vraw,
                    unsigned long long d_dm, unsigned long long d_cocc, unsigned long long d_exc)
{
    TailDev *c = (TailDev *)h;
    if (!c) return -1;
    c->err[0] = 0;
    if (!d_J || !d_vraw || !d_dm || !d_cocc || nhist < 1 || nhist > TL_SPACE || slot < 0 || slot >= TL_SPACE || !hist) {
        snprintf(c->err, sizeof c->err, "DFT_ScfTailStep: bad arguments");
        return -1;
    }
    TailArgs a{};
    a.n = c->n; a.no = c->no; a.slot = slot; a.nhist = nhist; a.have_coef = coef != nullptr; a.rotate = rotate != 0;
    a.max_inner = max_inner > 0 ? max_inner : 60;
    bool has_slot = false;
    for (int s = 0; s < nhist; ++s) {
        if (hist[s] < 0 || hist[s] >= TL_SPACE) { snprintf(c->err, sizeof c->err, "DFT_ScfTailStep: bad ring slot"); return -1; }
        a.hist[s] = hist[s];
        a.coef[s] = coef ? coef[s] : 0.0;
        has_slot |= hist[s] == slot;
    }
    if (!has_slot) { snprintf(c->err, sizeof c->err, "DFT_ScfTailStep: the new slot is not in the history"); return -1; }
    a.c_hf = c_hf; a.tol = tol; a.canon_tol = canon_tol > 0.0 ? canon_tol : 1e-3;
    c->last = a;
    const int n = c->n;
    const double *J = (const double *)d_J, *K = (const double *)d_K, *V = (const double *)d_vraw;
    double *dm = (double *)d_dm, *cocc = (double *)d_cocc;
    hipStream_t st = c->stream;
    const unsigned long seq = ++c->seq;
    const int nt = (n + 15) / 16;
    hipLaunchKernelGGL(k_tail_fock, dim3(n), dim3(256), 0, st, a, c->H, c->S, J, K, V, cocc, c->Fb, c->FC, c->SC, c->status);
    hipLaunchKernelGGL(k_tail_err, dim3(n), dim3(128), 0, st, a, c->FC, c->SC, c->Eb, c->gpart);
    hipLaunchKernelGGL(k_tail_mix, dim3(n), dim3(128), 0, st, a, c->gpart, c->Gb, c->Fb, c->Fx, c->status);
    if (a.rotate) {
        hipLaunchKernelGGL(k_tail_gemm, dim3(nt * nt), dim3(256), 0, st, n, n, n, c->Fx, n, 1, c->U, n, 1, c->FU, n, 1.0, 0, c->status);   // F_ext U
        hipLaunchKernelGGL(k_tail_gemm, dim3(nt * nt), dim3(256), 0, st, n, n, n, c->U, 1, n, c->FU, n, 1, c->A, n, 1.0, 0, c->status);    // U^T (F_ext U)
        long long *stamps = (long long *)(c->status + 8);
        if (!c->big) {
            hipLaunchKernelGGL(k_tail_rot, dim3(1), dim3(TL_ROT_T), c->rot_lds, st, a, c->lo, c->A, c->U, c->Km, c->Rm, c->Qm, c->Bm, c->FU,
                               c->eig, c->status, stamps);
        } else {
            hipLaunchKernelGGL(k_tail_rot_big, dim3(1), dim3(TL_ROT_T), c->rot_lds, st, a, c->A, c->Km, c->K2, c->Kt, c->Kt2, c->Qm, c->Rm, c->Bm,
                               c->rden, c->smalls, c->FU, c->eig, c->status, stamps, c->Kfix, c->KXg, 1, c->state);
            launch_big_steps(c, a, c->steps_hint);
            launch_big_finish(c, a);
        }
        hipLaunchKernelGGL(k_tail_gemm, dim3(nt * nt), dim3(256), 0, st, n, n, n, c->U, n, 1, c->FU, n, 1, c->Unew, n, 1.0, 0, c->status);   // U' = U W
    } else {
        hipLaunchKernelGGL(k_tail_need_exact, dim3(1), dim3(1), 0, st, c->status);
    }
    hipLaunchKernelGGL(k_tail_density, dim3(n), dim3(128), 0, st, a, 0, seq, c->H, J, K, c->U, c->Unew, dm, cocc, c->epart,
                       c->status, (const double *)d_exc, c->h_out_dev);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { tail_error(c, "SCF tail launch", e); return -1; }
    return 0;
}

// After status 3 (the memory-resident rotation ran out of queued fixed-point steps): `nsteps` more, then the rest of the step.
int DFT_ScfTailMore(void *h, int nsteps, unsigned long long d_J, unsigned long long d_K, unsigned long long d_dm, unsigned long long d_cocc,
                    unsigned long long d_exc)
{
    TailDev *c = (TailDev *)h;
    if (!c || !c->big || !d_J || !d_dm || !d_cocc || nsteps < 1) return -1;
    c->err[0] = 0;
    const TailArgs a = c->last;
    const int n = c->n, nt = (n + 15) / 16;
    hipStream_t st = c->stream;
    const unsigned long seq = ++c->seq;
    hipLaunchKernelGGL(k_tail_clear_more, dim3(1), dim3(1), 0, st, c->status);
    launch_big_steps(c, a, nsteps);
    launch_big_finish(c, a);
    hipLaunchKernelGGL(k_tail_gemm, dim3(nt * nt), dim3(256), 0, st, n, n, n, c->U, n, 1, c->FU, n, 1, c->Unew, n, 1.0, 0, c->status);
    hipLaunchKernelGGL(k_tail_density, dim3(n), dim3(128), 0, st, a, 0, seq, c->H, (const double *)d_J, (const double *)d_K, c->U, c->Unew,
                       (double *)d_dm, (double *)d_cocc, c->epart, c->status, (const double *)d_exc, c->h_out_dev);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { tail_error(c, "SCF tail launch", e); return -1; }
    return 0;
}

// Fixed-point steps queued per DFT_ScfTailStep on the memory-resident path (the caller knows how many the last cycle took)
int DFT_ScfTailSetStepsHint(void *h, int nsteps)
{
    TailDev *c = (TailDev *)h;
    if (!c) return -1;
    c->steps_hint = nsteps < 1 ? 1 : nsteps > 60 ? 60 : nsteps;
    return 0;
}

int DFT_ScfTailFinish(void *h, double c_hf, unsigned long long d_J, unsigned long long d_K, unsigned long long d_dm, unsigned long long d_cocc)
{
    TailDev *c = (TailDev *)h;
    if (!c) return -1;
    c->err[0] = 0;
    if (!d_J || !d_dm || !d_cocc) { snprintf(c->err, sizeof c->err, "DFT_ScfTailFinish: bad arguments"); return -1; }
    TailArgs a{};
    a.n = c->n; a.no = c->no; a.c_hf = c_hf;
    const unsigned long seq = ++c->seq;
    hipLaunchKernelGGL(k_tail_begin, dim3(1), dim3(1), 0, c->stream, c->status);
    hipLaunchKernelGGL(k_tail_density, dim3(c->n), dim3(128), 0, c->stream, a, 1, seq, c->H, (const double *)d_J, (const double *)d_K,
                       c->U, c->Unew, (double *)d_dm, (double *)d_cocc, c->epart, c->status, (const double *)nullptr, c->h_out_dev);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { tail_error(c, "SCF tail launch", e); return -1; }
    return 0;
}

int DFT_ScfTailWait(void *h, double *out)
{
    TailDev *c = (TailDev *)h;
    if (!c || !out) return -1;
    volatile unsigned long *word = (volatile unsigned long *)c->h_out + 8;
    for (unsigned spins = 1; *word != c->seq; ++spins) {
        if ((spins & 0xFFF) == 0) {
            const hipError_t q = hipStreamQuery(c->stream);
            if (q != hipErrorNotReady) {
                if (*word == c->seq) break;
                if (q != hipSuccess) { tail_error(c, "SCF tail", q); return -1; }
                // the stream is idle and the word has not arrived: one more look, then give up
                if (hipStreamSynchronize(c->stream) == hipSuccess && *word == c->seq) break;
                snprintf(c->err, sizeof c->err, "SCF tail: the result word never arrived");
                return -1;
            }
        }
        __builtin_ia32_pause();
    }
    for (int i = 0; i < 8; ++i) out[i] = ((volatile double *)c->h_out)[i];
    return 0;
}

// Diagnostics: wall-clock stamps (100 MHz) of the last rotation kernel's phases (tools/tail_time.py)
int DFT_ScfTailStamps(void *h, long long *host_out16)
{
    TailDev *c = (TailDev *)h;
    if (!c || !host_out16) return -1;
    if (hipStreamSynchronize(c->stream) != hipSuccess ||
        hipMemcpy(host_out16, c->status + 8, 16 * sizeof(long long), hipMemcpyDeviceToHost) != hipSuccess) { (void)hipGetLastError(); return -1; }
    return 0;
}

int DFT_ScfTailGram(void *h, double *host_out)
{
    TailDev *c = (TailDev *)h;
    if (!c || !host_out) return -1;
    const hipError_t e = hipMemcpyAsync(host_out, c->Gb, sizeof(double) * TL_SPACE * TL_SPACE, hipMemcpyDeviceToHost, c->stream);
    if (e != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess) { tail_error(c, "copy Gram matrix", e); return -1; }
    return 0;
}

} // extern "C"
