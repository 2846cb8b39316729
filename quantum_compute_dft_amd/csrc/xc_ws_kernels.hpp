// Wave-specialised contraction kernels for nao <= 128 (NT = ceil(nao/16) <= 8 MFMA column
// tiles): the sizes of every BASELINE config that fits dense ERIs (H2O/def2-SVP 24,
// Benzene/def2-SVP 114, sto-3g 7..80).  Same mathematics as xc_kernels.hpp (the generic
// path for nao > 128).
//
// Why this shape (measured on MI355X, profiles/r01_*): a lock-stepped workgroup alternates
// "HBM loads in flight" and "MFMA / epilogue", so neither pipe is busy more than ~60 % of
// the time (rho 124 us, vxc 129 us against an 83 us HBM floor for 4 AO planes).  Here one
// 512-thread workgroup per CU is split by role:
//   waves 0-3  MFMA waves   : fp64 MFMA only, operands from an LDS ring;
//   waves 4-7  loader waves : 16-byte coalesced plane loads (16 lanes = 256 contiguous bytes
//                             of one grid row; thread = (row, seg), columns 32j+2seg+{0,1}),
//                             two 16-point sub-tiles always in flight in registers, LDS
//                             staging, and the cheap VALU work (B rows / row dots).
// The roles meet at ONE s_barrier per sub-tile; the ring is deep enough that a stage is
// rewritten only after a later barrier than its last read, so no flags are needed:
//   vxc : loaders write stage i (sub-tile i) at step i, MFMA waves read it at step i+2.
//   rho : loaders write AO stage i at step i, MFMA waves form X(i) at step i+1 into an
//         X ring, loaders take the row dots of sub-tile i at step i+2.
// Persistent: workgroup b owns sub-tiles b, b+grid, ... (neighbouring CUs stream
// neighbouring HBM pages).  LDS leading dimensions: AO tile = 2 or 18 (mod 32) doubles (the
// 16-row x 2-k A-operand read hits 32 distinct bank pairs), X/P/Q tiles = 16 (mod 32)
// (2 k-rows x 16 columns likewise).
//
// References replaced: src/dft_solver.cu:294-307,346-380 (rho kernels),
// :309-513 pass-2 B rows + :541-548 cublasDgemm (Vxc).
#pragma once
#include <hip/hip_runtime.h>
#include "xc_kernels.hpp"

namespace qcdft {

#ifdef QCDFT_STAMPS // diagnostic build only (tools/ws_stamp_probe.hip): per-wave cycle shares
__device__ unsigned long long g_stamps[256 * 8 * 4];
#define QCDFT_T(var) unsigned long long var = __builtin_amdgcn_s_memtime()
#define QCDFT_ACC(slot, a, b) st_acc[slot] += (b) - (a)
#else
#define QCDFT_T(var)
#define QCDFT_ACC(slot, a, b)
#endif

constexpr int WS_ROWS = 16;      // grid points per sub-tile
constexpr int WS_THREADS = 512;  // 4 MFMA waves + 4 loader waves
constexpr int WS_RING = 4;       // LDS ring depth (static stage index under 4x unrolling)
constexpr int WS_OT = 16;        // sub-tiles of density results per burst of stores (k_rho_ws)

template <int NT> struct WsCfg {
    static constexpr int NCOL = 16 * NT;                      // padded AO columns
    static constexpr int JN = (NT + 1) / 2;                   // 32-column groups per row
    static constexpr int LDA = NCOL + 2;                      // = 2 or 18 (mod 32)
    static constexpr int LDX = ((NCOL + 31) / 32) * 32 + 16;  // = 16 (mod 32)
};

// Sub-tile of workgroup `b` at its s-th step; `rev` walks the grid from its END (the Vxc kernel does:
// it starts where the density kernel stopped, so the tail of the planes is still in the Infinity Cache:
// -5 % on either kernel, tools/ws_order.py).
__device__ __forceinline__ unsigned ws_tile(unsigned ntile, unsigned b, unsigned s, unsigned nwg, int rev)
{
    const unsigned t = b + s * nwg;
    return rev ? ntile - 1u - t : t;
}

// ------------------------------------------------------------------ Vxc ----
// V[a][b] += sum_g Q[g][a] P[g][b],  Q = sum_c coef_c * plane_c,  P = AO.
// SYM: the workgroup writes M + M^T of its partial (B3LYP, symmetrize_matrix_kernel
// src/dft_solver.cu:515-527) -- transposed through LDS so the slab reduce stays a coalesced sum.
template <int NT, bool GRAD, bool VEC, bool SYM>
__global__ __launch_bounds__(WS_THREADS, 2) void k_vxc_ws(long ngrid, int nao,
                                                          const double *__restrict__ ao,
                                                          const double *__restrict__ gx,
                                                          const double *__restrict__ gy,
                                                          const double *__restrict__ gz,
                                                          const double *__restrict__ coef,
                                                          double *__restrict__ slabs, int rev)
{
    using C = WsCfg<NT>;
    constexpr int TILE = WS_ROWS * C::LDX;
    constexpr int NTW = (NT + 1) / 2; // MFMA tiles per wave along each of a, b
    __shared__ double ring[2 * WS_RING * TILE];
    double *const Ps = ring, *const Qs = ring + WS_RING * TILE;

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const long ntile = (ngrid + WS_ROWS - 1) / WS_ROWS;
    // sub-tiles of this workgroup: blockIdx.x + k*gridDim.x, k < nloc
    const long nloc = (ntile > (long)blockIdx.x) ? (ntile - blockIdx.x + gridDim.x - 1) / gridDim.x : 0;
    const long nstep = nloc + 2; // ring latency of two steps

    if (wave < 4) {
        // ---------------------------------------------------------- MFMA role
        const int lane = tid & 63, li = lane & 15, lk = lane >> 4;
        const int wa = wave >> 1, wb = wave & 1;
        const int na = (NT - wa + 1) / 2, nb = (NT - wb + 1) / 2; // owned tiles wa+2i, wb+2j
        d4 acc[NTW][NTW];
#pragma unroll
        for (int i = 0; i < NTW; ++i)
#pragma unroll
            for (int j = 0; j < NTW; ++j) acc[i][j] = (d4){0.0, 0.0, 0.0, 0.0};

#ifdef QCDFT_STAMPS
        unsigned long long st_acc[4] = {0, 0, 0, 0};
#endif
        // Operand fragments are software-pipelined one k-step ahead, ACROSS the step barrier: the
        // slot consumed at step s+1 was published by the barrier of step s-1 (ring latency 2), so
        // its first fragments are fetched before this step's barrier and the MFMA pipe restarts
        // without an LDS round trip (the barrier only hands slots back to the loaders).
        const int fo = lk * C::LDX + li;
        double af[NTW], bf[NTW];
        auto load_frags = [&](int slot, int ks, double (&a_)[NTW], double (&b_)[NTW]) {
            const double *P = Ps + slot * TILE + fo + 4 * ks * C::LDX;
            const double *Q = Qs + slot * TILE + fo + 4 * ks * C::LDX;
#pragma unroll
            for (int i = 0; i < NTW; ++i) {
                a_[i] = Q[16 * min(wa + 2 * i, NT - 1)]; // clamped: unowned tiles skipped below
                b_[i] = P[16 * min(wb + 2 * i, NT - 1)];
            }
        };
#pragma unroll
        for (int i = 0; i < NTW; ++i) { af[i] = 0.0; bf[i] = 0.0; }
        for (long base = 0; base < nstep; base += WS_RING) {
#pragma unroll
            for (int u = 0; u < WS_RING; ++u) {
                const long step = base + u;
                QCDFT_T(ta);
                if (step >= 2 && step < nstep) { // consume sub-tile step-2 from stage (u+2)%4
                    if (step == 2) load_frags((u + 2) % WS_RING, 0, af, bf); // first consumed sub-tile
#pragma unroll
                    for (int ks = 0; ks < WS_ROWS / 4; ++ks) {
                        double an[NTW], bn[NTW];
                        if (ks + 1 < WS_ROWS / 4) load_frags((u + 2) % WS_RING, ks + 1, an, bn);
                        else                      load_frags((u + 3) % WS_RING, 0, an, bn); // next step's slot (may be unused garbage at the tail)
#pragma unroll
                        for (int i = 0; i < NTW; ++i)
#pragma unroll
                            for (int j = 0; j < NTW; ++j)
                                if ((2 * i + 1 < NT || i < na) && (2 * j + 1 < NT || j < nb))
                                    acc[i][j] = mfma_f64(af[i], bf[j], acc[i][j]);
#pragma unroll
                        for (int i = 0; i < NTW; ++i) { af[i] = an[i]; bf[i] = bn[i]; }
                    }
                }
                QCDFT_T(tb);
                __syncthreads();
                QCDFT_T(tc);
                QCDFT_ACC(0, ta, tb);
                QCDFT_ACC(1, tb, tc);
            }
        }
#ifdef QCDFT_STAMPS
        if ((tid & 63) == 0) { g_stamps[(blockIdx.x * 8 + wave) * 4 + 0] = st_acc[0]; g_stamps[(blockIdx.x * 8 + wave) * 4 + 1] = st_acc[1]; }
#endif
        double *slab = slabs + (size_t)blockIdx.x * nao * nao;
        if (!SYM) {
#pragma unroll
            for (int i = 0; i < NTW; ++i)
#pragma unroll
                for (int j = 0; j < NTW; ++j) {
                    if (i >= na || j >= nb) continue;
                    const int b = 16 * (wb + 2 * j) + li;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int a = 16 * (wa + 2 * i) + lk + 4 * r;
                        if (a < nao && b < nao) slab[(size_t)a * nao + b] = acc[i][j][r];
                    }
                }
        } else {
            // the ring is dead after the last barrier of the loop: reuse it as M[a][b], ld NCOL+1
            constexpr int LDM = C::NCOL + 1;
            static_assert(LDM * C::NCOL <= 2 * WS_RING * TILE, "M tile must fit in the ring");
            double *M = ring;
#pragma unroll
            for (int i = 0; i < NTW; ++i)
#pragma unroll
                for (int j = 0; j < NTW; ++j) {
                    if (i >= na || j >= nb) continue;
                    const int b = 16 * (wb + 2 * j) + li;
#pragma unroll
                    for (int r = 0; r < 4; ++r) M[(16 * (wa + 2 * i) + lk + 4 * r) * LDM + b] = acc[i][j][r];
                }
        }
    } else {
        // -------------------------------------------------------- loader role
        // The loader stream is short (~200 instructions per step) but latency-critical; at equal
        // priority the SIMD's arbiter (oldest first) lets the back-to-back MFMA wave starve it
        // (measured with s_memtime stamps: 7k cycles per step).  The MFMA wave needs one issue
        // slot per 64 cycles, so giving the loaders priority costs it nothing.
        __builtin_amdgcn_s_setprio(3);
        const int lt = tid - 256, row = lt >> 4, seg = lt & 15;
        const double *c0 = coef, *c1 = coef + (size_t)ngrid, *c2 = coef + 2 * (size_t)ngrid,
                     *c3 = coef + 3 * (size_t)ngrid;
        // two register sets: sub-tiles s and s+1 in flight
        double p0[2][2 * C::JN], p1[2][2 * C::JN], p2[2][2 * C::JN], p3[2][2 * C::JN];
        double k0[2], k1[2], k2[2], k3[2];

        // Loads are issued UNCONDITIONALLY (a drain step's tile offset is the plane size: out of range,
        // no traffic): with a conditional issue the number of outstanding loads is path-dependent and the
        // compiler falls back to `s_waitcnt vmcnt(0)`, which drains the younger register set as
        // well and collapses the prefetch to one step (measured: 7k cycles per step).
        const long plane = ngrid * (long)nao;
        const unsigned voff = (unsigned)(row * nao + 2 * seg) * 8u, koff = (unsigned)row * 8u;
        const unsigned tile_b = (unsigned)(WS_ROWS * nao) * 8u, ktile_b = WS_ROWS * 8u;
        const unsigned plane_b = (unsigned)(plane * 8), coef_b = (unsigned)(ngrid * 8);
        const __amdgpu_buffer_rsrc_t r0 = plane_rsrc(ao, plane), r1 = plane_rsrc(GRAD ? gx : ao, plane),
                                     r2 = plane_rsrc(GRAD ? gy : ao, plane), r3 = plane_rsrc(GRAD ? gz : ao, plane),
                                     q0 = plane_rsrc(c0, ngrid), q1 = plane_rsrc(GRAD ? c1 : c0, ngrid),
                                     q2 = plane_rsrc(GRAD ? c2 : c0, ngrid), q3 = plane_rsrc(GRAD ? c3 : c0, ngrid);
        auto issue = [&](int set, unsigned s) {
            const bool live = s < (unsigned)nloc;                         // drain steps load nothing
            const unsigned t = ws_tile((unsigned)ntile, blockIdx.x, s, gridDim.x, rev); // wave-uniform
            const unsigned so = live ? t * tile_b : plane_b, ko = live ? t * ktile_b : coef_b;
            k0[set] = buf_load_f64(q0, koff, ko);
            if (GRAD) {
                k1[set] = buf_load_f64(q1, koff, ko);
                k2[set] = buf_load_f64(q2, koff, ko);
                k3[set] = buf_load_f64(q3, koff, ko);
            }
            buf_load_row<C::JN, VEC>(r0, voff, so, p0[set]);
            if (GRAD) {
                buf_load_row<C::JN, VEC>(r1, voff, so, p1[set]);
                buf_load_row<C::JN, VEC>(r2, voff, so, p2[set]);
                buf_load_row<C::JN, VEC>(r3, voff, so, p3[set]);
            }
        };
        issue(0, 0);
        __builtin_amdgcn_sched_barrier(0); // program order: set 0 must be the OLDER one when the loop waits for it
        issue(1, 1);
        __builtin_amdgcn_sched_barrier(0);
#ifdef QCDFT_STAMPS
        unsigned long long st_acc[4] = {0, 0, 0, 0};
#endif

        for (long base = 0; base < nstep; base += WS_RING) {
#pragma unroll
            for (int u = 0; u < WS_RING; ++u) {
                const long step = base + u;
                const int set = u & 1;
                QCDFT_T(ta);
#ifdef QCDFT_STAMPS
                asm volatile("s_waitcnt vmcnt(20)" ::: "memory"); // the 20 loads of the newer set may stay in flight
#endif
                QCDFT_T(tw);
                { // stage sub-tile `step` into ring slot u (steps >= nloc stage rows that are never read;
                  // rows past the grid arrive as zeros from the range-checked loads)
                    double *P = Ps + u * TILE, *Q = Qs + u * TILE;
#pragma unroll
                    for (int j = 0; j < C::JN; ++j) {
                        const int c = 32 * j + 2 * seg;
                        if (c < C::NCOL) {
                            const double a0 = p0[set][2 * j], b0 = p0[set][2 * j + 1];
                            double qa = k0[set] * a0, qb = k0[set] * b0;
                            if (GRAD) {
                                qa += k1[set] * p1[set][2 * j] + k2[set] * p2[set][2 * j] + k3[set] * p3[set][2 * j];
                                qb += k1[set] * p1[set][2 * j + 1] + k2[set] * p2[set][2 * j + 1] + k3[set] * p3[set][2 * j + 1];
                            }
                            *reinterpret_cast<double2 *>(&Q[row * C::LDX + c]) = make_double2(qa, qb);
                            *reinterpret_cast<double2 *>(&P[row * C::LDX + c]) = make_double2(a0, b0);
                        }
                    }
                    QCDFT_T(tm);
                    issue(set, (unsigned)step + 2u);
                    QCDFT_ACC(3, tw, tm);
                }
                QCDFT_T(tb);
                __syncthreads();
                QCDFT_T(tc);
                QCDFT_ACC(0, ta, tw);
                QCDFT_ACC(1, tw, tb);
                QCDFT_ACC(2, tb, tc);
            }
        }
#ifdef QCDFT_STAMPS
        if ((tid & 63) == 0) { for (int q = 0; q < 4; ++q) g_stamps[(blockIdx.x * 8 + wave) * 4 + q] = st_acc[q]; }
#endif
    }
    if (SYM) { // all eight waves: slab = M + M^T, coalesced stores
        constexpr int LDM = C::NCOL + 1;
        const double *M = ring;
        __syncthreads();
        double *slab = slabs + (size_t)blockIdx.x * nao * nao;
        for (int e = tid; e < nao * nao; e += WS_THREADS) {
            const int a = e / nao, b = e - a * nao;
            slab[e] = M[a * LDM + b] + M[b * LDM + a]; // (x + y) == (y + x): bitwise symmetric
        }
    }
}

// ------------------------------------------------------------------ rho ----
// rho_g = sum_v X[g][v] AO[g][v],  grad rho_g = 2 sum_v X[g][v] dAO[g][v],  X = AO . Ds.
// MFMA wave w owns column tiles {w, w+4}: their slices of Ds stay in registers.
template <int NT, bool GRAD, bool VEC>
__global__ __launch_bounds__(WS_THREADS, 2) void k_rho_ws(long ngrid, int nao,
                                                          const double *__restrict__ ao,
                                                          const double *__restrict__ gx,
                                                          const double *__restrict__ gy,
                                                          const double *__restrict__ gz,
                                                          const double *__restrict__ dm,
                                                          double *__restrict__ rho,
                                                          double *__restrict__ grad,
                                                          double *__restrict__ sigma, int rev)
{
    using C = WsCfg<NT>;
    constexpr int NKS = 4 * NT;       // k-steps over the padded AO index
    constexpr int NTW = (NT + 3) / 4; // column tiles per MFMA wave
    constexpr int ATILE = WS_ROWS * C::LDA, XTILE = WS_ROWS * C::LDX;
    __shared__ double As[WS_RING * ATILE];
    __shared__ double Xs[2 * XTILE];
    // Results leave in BURSTS: rho / grad rho of WS_OT sub-tiles are collected here and written by the loader waves in
    // one go (40 bytes per grid row, 1 % of the traffic -- but stored sub-tile by sub-tile the trickle of writes keeps
    // turning the HBM channels around under the read stream: 10 us of 85 in tools/stream_pattern_probe4.hip).
    __shared__ double Ob[(WS_OT + 1) * WS_ROWS * 4];

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const long ntile = (ngrid + WS_ROWS - 1) / WS_ROWS;
    const long nloc = (ntile > (long)blockIdx.x) ? (ntile - blockIdx.x + gridDim.x - 1) / gridDim.x : 0;
    const long nstep = nloc + 2;

    if (wave < 4) {
        // ---------------------------------------------------------- MFMA role
        const int lane = tid & 63, li = lane & 15, lk = lane >> 4;
        const int nks = (nao + 3) >> 2; // k-steps that carry data
        const bool two = NTW > 1 && wave + 4 < NT;
        double dreg[NTW][NKS];
#pragma unroll
        for (int i = 0; i < NTW; ++i) {
            const int tcol = min(wave + 4 * i, NT - 1); // clamped: an unowned tile is never used
            // Ds = (D + D^T)/2, zero outside nao x nao: formed here, once per wave, straight from
            // the caller's matrix (saves the separate symmetrisation launch of the generic path)
            const int n = 16 * tcol + li;
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
                const int k = 4 * ks + lk;
                const bool in = k < nao && n < nao;
                const int kc = in ? k : 0, nc = in ? n : 0;
                const double v = 0.5 * (dm[(size_t)kc * nao + nc] + dm[(size_t)nc * nao + kc]);
                dreg[i][ks] = in ? v : 0.0;
            }
        }
        for (long base = 0; base < nstep; base += WS_RING) {
#pragma unroll
            for (int u = 0; u < WS_RING; ++u) {
                const long step = base + u;
                if (step >= 1 && step <= nloc && wave < NT) { // X(step-1) from AO stage (u+3)%4
                    const double *ap = As + ((u + 3) % WS_RING) * ATILE + li * C::LDA + lk;
                    double *X = Xs + ((u + 1) & 1) * XTILE;
                    d4 acc[NTW];
#pragma unroll
                    for (int i = 0; i < NTW; ++i) acc[i] = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
                    for (int ks = 0; ks < NKS; ++ks) {
                        if (ks < NKS - 3 || ks < nks) {
                            const double a = ap[4 * ks];
                            acc[0] = mfma_f64(a, dreg[0][ks], acc[0]);
                            if (NTW > 1) {
                                if (two) acc[NTW - 1] = mfma_f64(a, dreg[NTW - 1][ks], acc[NTW - 1]);
                            }
                        }
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        X[(lk + 4 * r) * C::LDX + 16 * wave + li] = acc[0][r];
                        if (NTW > 1) {
                            if (two) X[(lk + 4 * r) * C::LDX + 16 * (wave + 4) + li] = acc[NTW - 1][r];
                        }
                    }
                }
                __syncthreads();
            }
        }
    } else {
        // -------------------------------------------------------- loader role
        __builtin_amdgcn_s_setprio(3); // see k_vxc_ws
        const int lt = tid - 256, row = lt >> 4, seg = lt & 15;
        double ph[2][2 * C::JN];                                           // AO of sub-tiles s, s+1
        double pgx[2][2 * C::JN], pgy[2][2 * C::JN], pgz[2][2 * C::JN];   // gradients of s-2.., see below

        auto row_of = [&](long s) { return (long)ws_tile((unsigned)ntile, blockIdx.x, (unsigned)s, gridDim.x, rev) * WS_ROWS + row; };
        // unconditional issue, drain steps out of range: see k_vxc_ws
        const long plane = ngrid * (long)nao;
        const unsigned voff = (unsigned)(row * nao + 2 * seg) * 8u;
        const unsigned tile_b = (unsigned)(WS_ROWS * nao) * 8u, plane_b = (unsigned)(plane * 8);
        const __amdgpu_buffer_rsrc_t r0 = plane_rsrc(ao, plane), r1 = plane_rsrc(GRAD ? gx : ao, plane),
                                     r2 = plane_rsrc(GRAD ? gy : ao, plane), r3 = plane_rsrc(GRAD ? gz : ao, plane);
        auto tile_off = [&](unsigned s) {
            return s < (unsigned)nloc ? ws_tile((unsigned)ntile, blockIdx.x, s, gridDim.x, rev) * tile_b : plane_b;
        };
        auto issue_ao = [&](int set, unsigned s) { buf_load_row<C::JN, VEC>(r0, voff, tile_off(s), ph[set]); };
        auto issue_grad = [&](int set, unsigned s) {
            const unsigned so = tile_off(s);
            buf_load_row<C::JN, VEC>(r1, voff, so, pgx[set]);
            buf_load_row<C::JN, VEC>(r2, voff, so, pgy[set]);
            buf_load_row<C::JN, VEC>(r3, voff, so, pgz[set]);
        };
        // Prologue in the loop's own issue order (AO 0, gradients 0, AO 1, gradients 1; the gradient loads are
        // drain-type: out of range, no traffic, into registers the loop overwrites before it reads them),
        // pinned with scheduling barriers.  The wait the compiler puts at the loop header is the MINIMUM over
        // the entry path and the back edge of "loads younger than the set consumed first": a prologue that
        // issues fewer or reordered loads turned that wait into vmcnt(0) -- a drain of the whole prefetch
        // once per trip (round-1 ISA).
        issue_ao(0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (GRAD) issue_grad(0, (unsigned)nloc);
        __builtin_amdgcn_sched_barrier(0);
        issue_ao(1, 1);
        __builtin_amdgcn_sched_barrier(0);
        if (GRAD) issue_grad(1, (unsigned)nloc);
        __builtin_amdgcn_sched_barrier(0);

        long flushed = 0;
        auto flush = [&](long j0, long j1) {
            const long j = j0 + (lt >> 4);
            const int r = lt & 15;
            if (j < j1 && j < nloc) {
                const long g = (long)ws_tile((unsigned)ntile, blockIdx.x, (unsigned)j, gridDim.x, rev) * WS_ROWS + r;
                if (g < ngrid) {
                    const double *o = Ob + ((j % (WS_OT + 1)) * WS_ROWS + r) * 4;
                    rho[g] = o[0];
                    if (GRAD) {
                        const double ax = o[1], ay = o[2], az = o[3];
                        grad[3 * g + 0] = ax;
                        grad[3 * g + 1] = ay;
                        grad[3 * g + 2] = az;
                        sigma[g] = ax * ax + ay * ay + az * az;
                    }
                }
            }
        };
        for (long base = 0; base < nstep; base += WS_RING) {
#pragma unroll
            for (int u = 0; u < WS_RING; ++u) {
                const long step = base + u;
                const int set = u & 1;
                // (a) AO(step) -> ring slot u, then refill the register set with AO(step+2)
                {
                    double *A = As + u * ATILE;
#pragma unroll
                    for (int j = 0; j < C::JN; ++j) {
                        const int c = 32 * j + 2 * seg;
                        if (c < C::NCOL)
                            *reinterpret_cast<double2 *>(&A[row * C::LDA + c]) = make_double2(ph[set][2 * j], ph[set][2 * j + 1]);
                    }
                    issue_ao(set, (unsigned)step + 2u);
                }
                // (b) row dots of sub-tile step-2: X from the X ring, AO from ring slot (u+2)%4,
                //     gradients from register set `set` (loaded at step-2)
                { // for step < 2 this runs on never-written LDS; nothing is stored (row_ok false)
                    const bool in_range = step >= 2 && step - 2 < nloc;
                    const long g = row_of(in_range ? step - 2 : 0);
                    const bool row_ok = in_range && g < ngrid;
                    const double *A = As + ((u + 2) % WS_RING) * ATILE;
                    const double *X = Xs + (u & 1) * XTILE;
                    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
#pragma unroll
                    for (int j = 0; j < C::JN; ++j) {
                        const int c = 32 * j + 2 * seg;
                        if (c < C::NCOL) {
                            const double2 x = *reinterpret_cast<const double2 *>(&X[row * C::LDX + c]);
                            const double2 p = *reinterpret_cast<const double2 *>(&A[row * C::LDA + c]);
                            s0 += x.x * p.x + x.y * p.y;
                            if (GRAD) {
                                // X columns >= nao are exact zeros (zero-padded Ds); rows past the grid read as 0
                                s1 += x.x * pgx[set][2 * j] + x.y * pgx[set][2 * j + 1];
                                s2 += x.x * pgy[set][2 * j] + x.y * pgy[set][2 * j + 1];
                                s3 += x.x * pgz[set][2 * j] + x.y * pgz[set][2 * j + 1];
                            }
                        }
                    }
                    s0 = row16_sum(s0);
                    if (GRAD) {
                        s1 = row16_sum(s1);
                        s2 = row16_sum(s2);
                        s3 = row16_sum(s3);
                    }
                    (void)row_ok;
                    if (seg == 0 && in_range) { // slot of sub-tile step-2 in the result ring
                        double *o = Ob + (((step - 2) % (WS_OT + 1)) * WS_ROWS + row) * 4;
                        o[0] = s0;
                        if (GRAD) { o[1] = 2.0 * s1; o[2] = 2.0 * s2; o[3] = 2.0 * s3; }
                    }
                }
                // the results of sub-tiles [flushed, step-2) are complete (written before the previous barrier):
                // one burst per WS_OT of them, thread = (sub-tile, row)
                if (step - 2 - flushed == WS_OT) {
                    flush(flushed, step - 2);
                    flushed = step - 2;
                }
                // (c) gradients of sub-tile `step` into the set just freed (consumed at step+2)
                if (GRAD) issue_grad(set, (unsigned)step);
                __syncthreads();
            }
        }
        flush(flushed, nloc); // the rest (at most WS_OT sub-tiles; the loop's last barrier made them visible)
    }
}

// Last launch of every DFT_ComputeXC call (one block): fixed-order sum of the per-block partials
// of k_xc_points (reduce_sum_kernel, src/dft_solver.cu:285-292, made deterministic), stored to the
// device scalar and, if given, to host-mapped memory.  Stream order puts it after every Vxc store
// of the call, so the host may return as soon as it sees the value (no copy launch, no sleeping
// synchronise).  (A last-block ticket with a fence inside the reduce kernel was tried first: its per-block
// __threadfence() cost 10-30 us.)
__global__ __launch_bounds__(256) void k_finish_exc(long npart, const double *__restrict__ partial,
                                                    double *__restrict__ exc_dev, double *exc_host)
{
    // strided partial sums (every thread's loads in flight together), a shuffle tree per wave, four wave sums through LDS:
    // one barrier.  (The launch is on the critical path of every synchronous call: an eight-barrier LDS tree took 4.0 us,
    // a single wave walking the partials one after another 5.8 us, rocprofv3.)
    __shared__ double part[4];
    double x = 0.0;
    for (long i = threadIdx.x; i < npart; i += 256) x += partial[i];
    for (int m = 32; m >= 1; m >>= 1) x += __shfl_down(x, m, 64);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = x;
    __syncthreads();
    if (threadIdx.x == 0) {
        const double e = (part[0] + part[1]) + (part[2] + part[3]);
        exc_dev[0] = e;
        if (exc_host) {
            *(volatile double *)exc_host = e;
            __threadfence_system();
        }
    }
}

// The call's last launch on the default path: the device scalar (complete: an earlier kernel of the stream wrote it) goes to
// the host-mapped word.  Seeing it, the host knows every kernel of the call has completed.
__global__ void k_publish_exc(const double *__restrict__ exc_dev, double *exc_host)
{
    *(volatile double *)exc_host = exc_dev[0];
    __threadfence_system();
}

// V = sum of the per-workgroup slabs in a fixed order (bitwise reproducible): 32 elements x 8
// slab groups per block, then a fixed tree over the groups.  SYM adds the transpose with
// transposed reads (validation path only; the production paths symmetrise in-kernel or with
// k_symmetrize).
// FIN: the highest-index block also does k_finish_exc's job (see the comment at the end of the kernel), saving
// that launch and its dispatch gap.
template <bool SYM, bool FIN = false>
__global__ __launch_bounds__(256) void k_reduce_slabs8(int nao, int nslab,
                                                       const double *__restrict__ slabs,
                                                       double *__restrict__ V,
                                                       long npart = 0, const double *__restrict__ partial = nullptr,
                                                       double *__restrict__ exc_dev = nullptr, double *exc_host = nullptr)
{
    __shared__ double part[256];
    const size_t n2 = (size_t)nao * nao;
    const int el = threadIdx.x & 31, grp = threadIdx.x >> 5;
    const size_t e = (size_t)blockIdx.x * 32 + el;
    double s = 0.0;
    if (e < n2) {
        const int a = (int)(e / nao), b = (int)(e - (size_t)a * nao);
        const size_t et = (size_t)b * nao + a;
        // 16 slab loads in flight per thread (the slabs were written a moment ago and sit in L2 / the
        // Infinity Cache: the sum is latency-bound, 4 in flight took 15.6 us for 26.6 MB), summed in slab
        // order whatever the unrolling: bitwise reproducible
        int k = grp;
        for (; k + 8 * 15 < nslab; k += 8 * 16) {
            double v[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                v[q] = slabs[(size_t)(k + 8 * q) * n2 + e];
                if (SYM) v[q] += slabs[(size_t)(k + 8 * q) * n2 + et]; // (x + y) == (y + x): V comes out bitwise symmetric
            }
#pragma unroll
            for (int q = 0; q < 16; ++q) s += v[q];
        }
        for (; k < nslab; k += 8) {
            double v = slabs[(size_t)k * n2 + e];
            if (SYM) v += slabs[(size_t)k * n2 + et];
            s += v;
        }
    }
    part[threadIdx.x] = s;
    __syncthreads();
    if (grp == 0 && e < n2) {
        const double *p = &part[el];
        V[e] = ((p[0] + p[32]) + (p[64] + p[96])) + ((p[128] + p[160]) + (p[192] + p[224]));
    }
    if (FIN) {
        // The block with the HIGHEST index (dispatched last) also sums the Exc partials of k_xc_points -- an
        // earlier kernel of the call, complete by stream order -- into the device scalar and the host-mapped
        // word.  No ticket: 406 blocks drawing tickets on one address serialise at ~12 ns each, which cost
        // more (15.0 us) than a separate finishing launch (12.7 us); this form costs neither.  The word
        // therefore says "Exc is final and the call's last kernel is finishing" -- consumers on the solver's
        // stream are ordered behind it, others use option strict_sync (include/dft_solver.h).
        if (blockIdx.x == gridDim.x - 1) {
            __syncthreads();
            double x = 0.0;
            for (long i = threadIdx.x; i < npart; i += 256) x += partial[i];
            part[threadIdx.x] = x;
            __syncthreads();
            for (int m = 128; m >= 1; m >>= 1) {
                if ((int)threadIdx.x < m) part[threadIdx.x] += part[threadIdx.x + m];
                __syncthreads();
            }
            if (threadIdx.x == 0) {
                exc_dev[0] = part[0];
                if (exc_host) {
                    *(volatile double *)exc_host = part[0];
                    __threadfence_system();
                }
            }
        }
    }
}

// V = M + M^T (B3LYP, symmetrize_matrix_kernel src/dft_solver.cu:515-527) for the nao > 128 path:
// 32x32 tiles through LDS, block (ti <= tj) owns the tile pair, coalesced on both sides.
__global__ __launch_bounds__(256) void k_symmetrize(int nao, const double *__restrict__ M,
                                                    double *__restrict__ V)
{
    __shared__ double A[32][33], B[32][33];
    const int nt = (nao + 31) / 32;
    int ti = 0, rem = blockIdx.x; // linear block index -> (ti <= tj)
    while (rem >= nt - ti) { rem -= nt - ti; ++ti; }
    const int tj = ti + rem;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int r = ty; r < 32; r += 8) {
        const int a = 32 * ti + r, b = 32 * tj + tx;
        A[r][tx] = (a < nao && b < nao) ? M[(size_t)a * nao + b] : 0.0;
        const int a2 = 32 * tj + r, b2 = 32 * ti + tx;
        B[r][tx] = (a2 < nao && b2 < nao) ? M[(size_t)a2 * nao + b2] : 0.0;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int a = 32 * ti + r, b = 32 * tj + tx;
        if (a < nao && b < nao) V[(size_t)a * nao + b] = A[r][tx] + B[tx][r];
        const int a2 = 32 * tj + r, b2 = 32 * ti + tx;
        if (ti != tj && a2 < nao && b2 < nao) V[(size_t)a2 * nao + b2] = A[tx][r] + B[r][tx]; // same two addends: bitwise symmetric
    }
}

} // namespace qcdft
