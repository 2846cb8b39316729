// J and K from a factorised ERI,  (ij|kl) ~= sum_P L[P][i][j] L[P][k][l]   (SURVEY 8(f4)).
//
// The reference contracts the dense (nao^2, nao^2) ERI (J: dft_solver.cu:550-555, K: the einsum at
// dft.py:217-221), 8 nao^4 bytes that stop fitting in 288 GB at nao ~ 430.  With Cholesky vectors
// L (naux, nao, nao) the same matrices are
//     J      = sum_P (L_P : D) L_P                                  two HBM passes over L
//     Yt_P   = Cocc^T L_P                    (nocc x nao)            fp64 MFMA, 2 naux nao^2 nocc flop
//     K      = sum_P Yt_P^T Yt_P = Yt^T Yt,  Yt ((naux nocc) x nao)  fp64 MFMA, 2 naux nao^2 nocc flop
// with D = Cocc Cocc^T (occupation folded into Cocc).  This is the only formulation in which the
// exchange build is matrix-core work.
//
// Both MFMA steps are the same "TN" tile GEMM, C = A^T B with the contraction index on the rows of
// both row-major operands: 512 threads = 8 waves, BK = 16, LDS double-buffered, the next stage's
// range-checked buffer loads issued before the current stage's MFMAs (same skeleton as
// k_vxc_big).  Tile 128 x 256 (wave tile 64 x 64) for the Yt^T Yt step, split over the contraction
// index across workgroups of one XCD group with per-chunk slabs summed in fixed order; tile
// 64 x 256 (wave tile 16*MI x 32) for the half transform when nocc <= 64.
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>
#include "device_util.hpp"
#include "xc_big_kernels.hpp"

namespace qcdft {

constexpr int CD_BN = 256, CD_BK = 16, CD_LDB = CD_BN + 16; // 272 = 16 (mod 32)

// C[batch][chunk](M x N) = sum_{g in chunk} A[batch][g][a] * B[batch][g][b]
//   WGM = 2: tile 128 x 256, MI = 4.   WGM = 1: tile 64 x 256, MI = ceil(min(M,64)/16) row tiles live.
//   split = true : blockIdx.x -> (xcd = b % 8, slot = b / 8), pair = slot % npair, chunk = xcd + 8 (slot / npair), batch 0
//   split = false: pair = b % npair, batch = b / npair, one chunk covering [0, G)
// Rows of A/B past the chunk end read as zeros (descriptor range), columns past M/N only feed
// discarded outputs.
// DOT (half transform only, split = false): v_P = L_P : D is needed for J; with D = Cocc Cocc^T it equals
// sum_{i,b} Yt_P[i][b] Cocc[b][i], a dot of each RESULT tile with a block of the A operand, taken at the epilogue:
// one partial per (P, tile) in vpart -- the first of J's two passes over L costs no extra memory traffic.
//
// NW = waves per workgroup.  8: the tiles above, one workgroup per CU (LDS 90-106 KB).  4 (WGM = 1
// only): tile 64 x 128, 57 KB, TWO workgroups per CU whose prologues, barriers and epilogues
// interleave -- the half transform's contraction is only nao/16 stages long, so a lone workgroup
// per CU spends a tenth of its life filling and draining.
// BK = contraction rows per stage, NJ = 16-column MFMA tiles per wave (0 = the defaults above).  The half
// transform at nocc <= 64 runs <1, MI, ., ., DOT, 4, 8, 4>: wave tile 16 MI x 64, workgroup tile 64 x 256, 8-row
// stages.  Its MFMA waves also do their own staging, and next to an MFMA in flight a SIMD issues only ~2.6
// vector/LDS/VMEM (4 scalar) instructions per 64 cycles (tools/coissue_probe*.hip), so what sets the pace is
// the NON-MFMA instruction count per MFMA: the 16 x 32 wave tile needed 5 LDS reads per 6 MFMAs and ~100
// other instructions per 24-MFMA stage (57 % MFMA-busy, profiles/r01_pmc_kbuild.json); this one needs 7 reads
// per 12 MFMAs and ~45 per stage, and at 45 KB of LDS and <= 168 VGPRs (launch bound) THREE workgroups share a CU
// for MI <= 3 (nocc <= 48); MI = 4 (nocc 49-64: 64 accumulator VGPRs more) does not fit 168 registers -- it spilled
// 10-75 of them under that bound (round 2) -- and runs two workgroups per CU.
template <int WGM, int MI, bool VECA, bool VECB, bool DOT = false, int NW = 8, int BK_ = 0, int NJ_ = 0>
__global__ __launch_bounds__(64 * NW, (NW == 4 && BK_ == 8 && MI <= 3) ? 3 : 2) void k_gemm_tn(long G, int M, int N, int lda, int ldb,
                                                           const double *__restrict__ A, long strideA,
                                                           const double *__restrict__ B, long strideB,
                                                           long chunk, int nB, int npair, int split,
                                                           double *__restrict__ C, int ldc, long strideC_batch,
                                                           long strideC_chunk,
                                                           double *__restrict__ vpart = nullptr,
                                                           int skip_lower = 0)
{
    constexpr int BK = BK_ ? BK_ : CD_BK;
    constexpr int THREADS = 64 * NW, CG = THREADS / BK;                  // BK staging rows x CG column groups
    constexpr int BM = 64 * WGM, WGN = NW / WGM, NJ = NJ_ ? NJ_ : (WGM == 2 ? 4 : 2);   // wave tile 16 MI x 16 NJ
    constexpr int BN = 16 * NJ * WGN;                                     // 256 (NW 8) or 128 (NW 4)
    constexpr int LDA_ = BM + 16, LDB_ = BN + 16;                         // = 16 (mod 32)
    constexpr int ASZ = BK * LDA_, BSZ = BK * LDB_;
    constexpr int AH = BM / CG / 2;                                       // double2 loads per thread, A tile
    static_assert(BN / CG == 8, "B tile: 8 doubles per thread");
    __shared__ double lds[2 * (ASZ + BSZ)];
    double *const As = lds, *const Bs = lds + 2 * ASZ;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lk = lane >> 4;
    const int wm = wave / WGN, wn = wave % WGN;
    int pair, ck;
    long batch;
    if (split) {
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        pair = slot % npair;
        ck = xcd + 8 * (slot / npair);
        batch = 0;
    } else {
        pair = blockIdx.x % npair;
        batch = blockIdx.x / npair;
        ck = 0;
    }
    int a0 = (pair / nB) * BM, b0 = (pair % nB) * BN;
    if (skip_lower) {
        // C = A^T A (the exchange step): a tile whose rows all lie below its columns' block is the transpose
        // of a tile that is computed, so only the others are enumerated (`npair` counts them; the skipped
        // ones are filled in by k_mirror_lower after the slab sum).  Enumerated, not launched-and-exited:
        // workgroups go to CUs round-robin, and dead ones would leave the same CUs idle in every round.
        const int nA = (M + BM - 1) / BM;
        int cnt = 0;
        for (int ia = 0; ia < nA; ++ia)
            for (int ib = 0; ib < nB; ++ib)
                if (BM * ia < BN * ib + BN) {
                    if (cnt == pair) { a0 = BM * ia; b0 = BN * ib; }
                    ++cnt;
                }
    }
    const long glo = (long)ck * chunk, ghi = min(G, glo + chunk);
    A += batch * strideA;
    B += batch * strideB;

    d4 acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = (d4){0.0, 0.0, 0.0, 0.0};

    if (glo < ghi) {
        const int nst = (int)((ghi - glo + BK - 1) / BK);
        const int s_row = tid / CG, s_cq = tid % CG;
        const unsigned a_voff = (unsigned)(s_row * lda + a0 + 2 * AH * s_cq) * 8u;
        const unsigned b_voff = (unsigned)(s_row * ldb + b0 + 8 * s_cq) * 8u;
        // Two register sets: the loads of stage st+2 are issued before the MFMAs of stage st, a full
        // stage earlier than they are needed (one stage of the half transform is only ~0.7 us of MFMA,
        // less than an HBM round trip under load: with a one-stage prefetch its MFMA pipe was busy
        // 57 % of the time).  Every fetch issues the same number of loads -- stages past the end go
        // through a zero-record descriptor -- so the waits stay counted.
        double2 ra[2][AH], rb[2][4];
        // one descriptor per operand for the whole chunk, the stage selected by the SGPR offset of the loads
        // (range-checked by the hardware; a stage past the end passes the chunk size: every lane out of range,
        // no traffic, the loads still count in vmcnt).  The host keeps a chunk of either operand below 4 GiB.
        const __amdgpu_buffer_rsrc_t da = plane_rsrc(A + glo * lda, (ghi - glo) * (long)lda);
        const __amdgpu_buffer_rsrc_t db = plane_rsrc(B + glo * ldb, (ghi - glo) * (long)ldb);
        const unsigned a_end = (unsigned)((ghi - glo) * (long)lda * 8), b_end = (unsigned)((ghi - glo) * (long)ldb * 8);
        const unsigned a_step = (unsigned)(BK * lda) * 8u, b_step = (unsigned)(BK * ldb) * 8u;
        auto fetch = [&](auto S, int st) {
            constexpr int s = decltype(S)::value;
            const bool live = st < nst;
            const unsigned ao = live ? (unsigned)st * a_step : a_end, bo = live ? (unsigned)st * b_step : b_end;
#pragma unroll
            for (int h = 0; h < AH; ++h) ra[s][h] = buf_load_pair2<VECA>(da, a_voff + 16 * h, ao);
#pragma unroll
            for (int h = 0; h < 4; ++h) rb[s][h] = buf_load_pair2<VECB>(db, b_voff + 16 * h, bo);
        };
        auto stash = [&](auto S) { // register set s -> LDS buffer s
            constexpr int s = decltype(S)::value;
            double *Ad = As + s * ASZ + s_row * LDA_ + 2 * AH * s_cq;
#pragma unroll
            for (int h = 0; h < AH; ++h) *reinterpret_cast<double2 *>(Ad + 2 * h) = ra[s][h];
            double *Bd = Bs + s * BSZ + s_row * LDB_ + 8 * s_cq;
#pragma unroll
            for (int h = 0; h < 4; ++h) *reinterpret_cast<double2 *>(Bd + 2 * h) = rb[s][h];
        };
        auto compute = [&](int buf) {
            const double *Ap = As + buf * ASZ + lk * LDA_ + wm * 64 + li;
            const double *Bp = Bs + buf * BSZ + lk * LDB_ + wn * (16 * NJ) + li;
#pragma unroll
            for (int ks = 0; ks < BK / 4; ++ks) {
                double af[MI], bf[NJ];
#pragma unroll
                for (int i = 0; i < MI; ++i) af[i] = Ap[4 * ks * LDA_ + 16 * i];
#pragma unroll
                for (int j = 0; j < NJ; ++j) bf[j] = Bp[4 * ks * LDB_ + 16 * j];
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NJ; ++j) acc[i][j] = mfma_f64(af[i], bf[j], acc[i][j]);
            }
        };
        using S0 = std::integral_constant<int, 0>;
        using S1 = std::integral_constant<int, 1>;
        fetch(S0{}, 0);
        fetch(S1{}, 1);
        stash(S0{});
        lds_barrier();
        for (int st = 0; st < nst; st += 2) {
            // LDS buffer 0 holds stage st, register set 1 stage st+1
            fetch(S0{}, st + 2);
            compute(0);
            stash(S1{});
            lds_barrier();
            if (st + 1 < nst) { // uniform
                fetch(S1{}, st + 3);
                compute(1);
                stash(S0{});
                lds_barrier();
            }
        }
    }

    if (DOT) {
        // v_P = L_P : D with D = Cocc Cocc^T is  sum_{i,b} Yt_P[i][b] Cocc[b][i]: a dot of this tile's RESULT with
        // the (b, i) block of the A operand -- nocc x nao products at the epilogue instead of nao^2 products (and a
        // second load stream of D tiles) inside the stage loop.  One partial per (P, b-block), fixed order.
        double dot = 0.0;
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int b = b0 + wn * (16 * NJ) + 16 * j + li;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int a = a0 + wm * 64 + 16 * i + lk + 4 * r;
                    if (a < M && b < N) dot += acc[i][j][r] * A[(size_t)b * lda + a];
                }
            }
        __syncthreads(); // every wave is done with the tile buffers
        lds[tid] = dot;
        __syncthreads();
        for (int w = THREADS / 2; w > 0; w >>= 1) {
            if (tid < w) lds[tid] += lds[tid + w];
            __syncthreads();
        }
        if (tid == 0) vpart[(batch * (long)npair) + pair] = lds[0];
    }

    double *out = C + batch * strideC_batch + (long)ck * strideC_chunk;
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int b = b0 + wn * (16 * NJ) + 16 * j + li;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int a = a0 + wm * 64 + 16 * i + lk + 4 * r;
                if (a < M && b < N) out[(size_t)a * ldc + b] = acc[i][j][r];
            }
        }
}

// Cp[nu][i] = Cocc[nu][i] for i < nocc, 0 up to ldp (multiple of 16): aligned, zero-padded A operand
__global__ __launch_bounds__(256) void k_pack_cocc(int nao, int nocc, int ldp, const double *__restrict__ c,
                                                   double *__restrict__ cp)
{
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= (long)nao * ldp) return;
    const int r = (int)(e / ldp), i = (int)(e % ldp);
    cp[e] = i < nocc ? c[(size_t)r * nocc + i] : 0.0;
}

// (Reading only the upper triangle of the symmetric L_P against D + D^T halves the bytes but was slower at Benzene's
// size -- 112 against 89 us for the J step: the index arithmetic and half-idle waves cost more than the bytes save.)
// v[P] = sum_e L[P][e] D[e]   (one workgroup per vector, fixed summation order)
__global__ __launch_bounds__(256) void k_cd_dot(long n2, const double *__restrict__ L,
                                                const double *__restrict__ D, double *__restrict__ v)
{
    __shared__ double red[256];
    const double *Lp = L + (size_t)blockIdx.x * n2;
    double s = 0.0;
    for (long e = threadIdx.x; e < n2; e += 256) s += Lp[e] * D[e];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if (threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) v[blockIdx.x] = red[0];
}

// K[a][b] = K[b][a] for the tiles k_gemm_tn skipped (rows' 128-block entirely below the columns' 256-block)
__global__ __launch_bounds__(256) void k_mirror_lower(int n, double *__restrict__ K)
{
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= (long)n * n) return;
    const int a = (int)(e / n), b = (int)(e % n);
    if ((a / 128) * 128 >= (b / CD_BN) * CD_BN + CD_BN) K[e] = K[(size_t)b * n + a];
}

// v[P] = sum_b vpart[P][b]: the per-b-block partials the half transform leaves (DOT), fixed order -- unless the
// caller's dm turned out NOT to be cocc cocc^T (flag set by k_dm_consistency): then v[P] = L_P : dm as k_cd_dot
// computed it from dm itself.
__global__ __launch_bounds__(256) void k_cd_vsum(int naux, int nB, const double *__restrict__ vpart,
                                                 double *__restrict__ v, const int *__restrict__ flag = nullptr,
                                                 const double *__restrict__ vdot = nullptr)
{
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= naux) return;
    if (flag && *flag) {
        v[p] = vdot[p];
        return;
    }
    double s = 0.0;
    for (int b = 0; b < nB; ++b) s += vpart[(size_t)p * nB + b];
    v[p] = s;
}

// flag = 1 when dm differs from cocc cocc^T beyond round-off somewhere (a damped or mixed density, fractional
// occupations, orbitals passed without the sqrt(2)): J must then contract dm itself.  The host clears `flag`.
__global__ __launch_bounds__(256) void k_dm_consistency(int nao, int nocc, const double *__restrict__ dm,
                                                        const double *__restrict__ c, int *__restrict__ flag)
{
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= (long)nao * nao) return;
    const int a = (int)(e / nao), b = (int)(e % nao);
    const double *ca = c + (size_t)a * nocc, *cb = c + (size_t)b * nocc;
    double s = 0.0, m = 0.0;
    for (int i = 0; i < nocc; ++i) { s += ca[i] * cb[i]; m += fabs(ca[i] * cb[i]); }
    if (fabs(dm[e] - s) > 1e-11 * (m + fabs(dm[e])) + 1e-14) atomicOr(flag, 1);
}

// k_cd_dot that only works when the flag is set (a launch that exits at once otherwise)
__global__ __launch_bounds__(256) void k_cd_dot_if(const int *__restrict__ flag, long n2, const double *__restrict__ L,
                                                   const double *__restrict__ D, double *__restrict__ v)
{
    if (!*flag) return;
    __shared__ double red[256];
    const double *Lp = L + (size_t)blockIdx.x * n2;
    double s = 0.0;
    for (long e = threadIdx.x; e < n2; e += 256) s += Lp[e] * D[e];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if (threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) v[blockIdx.x] = red[0];
}

// part[y][e] = sum_{P in slice y} v[P] L[P][e]; slices of `pslice` vectors, summed afterwards
__global__ __launch_bounds__(256) void k_cd_axpy(long n2, int naux, int pslice, const double *__restrict__ L,
                                                 const double *__restrict__ v, double *__restrict__ part, int n = 0)
{
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= n2) return;
    const int p0 = blockIdx.y * pslice, p1 = min(naux, p0 + pslice);
    double s = 0.0;
    // every L_P is symmetric (include/dft_solver.h): only elements on or above the diagonal are read -- this second
    // pass over the vectors moves half their bytes -- and k_sym_from_upper fills the rest of J afterwards
    if (n > 0 && (int)(e % n) < (int)(e / n)) {
        part[(size_t)blockIdx.y * n2 + e] = 0.0;
        return;
    }
#pragma unroll 4
    for (int p = p0; p < p1; ++p) s += v[p] * L[(size_t)p * n2 + e];
    part[(size_t)blockIdx.y * n2 + e] = s;
}

// J[a][b] = J[b][a] for b < a (the axpy pass only produced the upper triangle)
__global__ __launch_bounds__(256) void k_sym_from_upper(int n, double *__restrict__ J)
{
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= (long)n * n) return;
    const int a = (int)(e / n), b = (int)(e % n);
    if (b < a) J[e] = J[(size_t)b * n + a];
}

} // namespace qcdft
