// Sixteen-wave form of the wave-specialised contraction kernels (nao <= 128): 1024-thread workgroups,
// one per CU, waves 0-7 issue fp64 MFMAs only, waves 8-15 stream the planes and do the vector work.
//
// Why sixteen: next to a wave that issues v_mfma_f64_16x16x4 back to back, EVERY other wave of that
// SIMD gets one vector/LDS/VMEM issue per ~25-40 cycles, whatever its priority -- but that budget is
// per WAVE, not per SIMD: two such waves issue twice as much, three 2.9x, and the MFMA wave stays at
// 64.0 cycles per MFMA throughout (tools/coissue_probe2.hip, profiles/r02_coissue_probe2.txt).  The
// eight-wave kernels (xc_ws_kernels.hpp) have ONE loader wave per SIMD whose ~150-200 instructions per
// 16-point sub-tile need more than the ~160 slots it gets during the 64 MFMAs of that sub-tile, so the
// loaders, not HBM and not the matrix pipe, set the pace (both pipes ~70 % busy).  Here each SIMD hosts
// two MFMA waves (half the accumulators each: 64 VGPRs, so sixteen waves fit the 128-VGPR budget) and
// two loader waves (thread = (row, seg) with 32 segments per grid row: half the loads, FMAs and LDS
// writes per wave).  Same ring, same single s_barrier per sub-tile.  X and the Vxc accumulation run in
// the same order per element as in the eight-wave kernels; the row dots are split over 32 instead of 16
// lanes, so rho (and everything downstream) differs from the eight-wave path in the last bits.
//
// References replaced: as xc_ws_kernels.hpp (src/dft_solver.cu:294-307,346-380; :309-513 pass 2 + :541-548).
#pragma once
#include <hip/hip_runtime.h>
#include "xc_ws_kernels.hpp"

namespace qcdft {

constexpr int W16_THREADS = 1024; // 8 MFMA waves + 8 loader waves

template <int NT> struct W16Cfg {
    static constexpr int NCOL = 16 * NT;                      // padded AO columns
    static constexpr int JN = (NT + 3) / 4;                   // 64-column groups per row
    static constexpr int LDA = NCOL + 2;                      // = 2 or 18 (mod 32)
    static constexpr int LDX = ((NCOL + 31) / 32) * 32 + 16;  // = 16 (mod 32)
};

// Sum over the 32 lanes of a half-wave (two DPP rows): 16-lane rotations, then row 0 -> row 1 and
// row 2 -> row 3 with row_bcast:15.  The total is valid in lanes 16-31 and 48-63.
__device__ __forceinline__ double half32_sum_hi(double v)
{
    v = row16_sum(v);
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x142, 0xA, 0xF, false); // row_bcast:15, rows 1 and 3
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x142, 0xA, 0xF, false);
    return v + __hiloint2double(hi, lo);
}

// ------------------------------------------------------------------ Vxc ----
// V[a][b] += sum_g Q[g][a] P[g][b],  Q = sum_c coef_c * plane_c,  P = AO.
// MFMA wave w = (wa = w>>1, wb = w&1) owns tiles a in {wa, wa+4} x b in {wb, wb+2, wb+4, wb+6}.
template <int NT, bool GRAD, bool VEC, bool SYM>
__global__ __launch_bounds__(W16_THREADS) void k_vxc_ws16(long ngrid, int nao,
                                                         const double *__restrict__ ao,
                                                         const double *__restrict__ gx,
                                                         const double *__restrict__ gy,
                                                         const double *__restrict__ gz,
                                                         const double *__restrict__ coef,
                                                         double *__restrict__ slabs, int dbg)
{
    using C = W16Cfg<NT>;
    constexpr int TILE = WS_ROWS * C::LDX;
    constexpr int NA = (NT + 3) / 4, NB = (NT + 1) / 2; // tiles per wave along a, b
    __shared__ double ring[2 * WS_RING * TILE];
    double *const Ps = ring, *const Qs = ring + WS_RING * TILE;

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const long ntile = (ngrid + WS_ROWS - 1) / WS_ROWS;
    const long nloc = (ntile > (long)blockIdx.x) ? (ntile - blockIdx.x + gridDim.x - 1) / gridDim.x : 0;
    const long nstep = nloc + 2; // ring latency of two steps
    const int rev = (dbg >> 16) & 1;

    if (wave < 8) {
        // ---------------------------------------------------------- MFMA role
        const int lane = tid & 63, li = lane & 15, lk = lane >> 4;
        const int wa = wave >> 1, wb = wave & 1;
        d4 acc[NA][NB];
#pragma unroll
        for (int i = 0; i < NA; ++i)
#pragma unroll
            for (int j = 0; j < NB; ++j) acc[i][j] = (d4){0.0, 0.0, 0.0, 0.0};
        // fragments one k-step ahead, across the step barrier (see k_vxc_ws)
        const int fo = lk * C::LDX + li;
        double af[NA], bf[NB];
        auto load_frags = [&](int slot, int ks, double (&a_)[NA], double (&b_)[NB]) {
            const double *P = Ps + slot * TILE + fo + 4 * ks * C::LDX;
            const double *Q = Qs + slot * TILE + fo + 4 * ks * C::LDX;
#pragma unroll
            for (int i = 0; i < NA; ++i) a_[i] = Q[16 * min(wa + 4 * i, NT - 1)]; // clamped: unowned tiles skipped below
#pragma unroll
            for (int j = 0; j < NB; ++j) b_[j] = P[16 * min(wb + 2 * j, NT - 1)];
        };
#pragma unroll
        for (int i = 0; i < NA; ++i) af[i] = 0.0;
#pragma unroll
        for (int j = 0; j < NB; ++j) bf[j] = 0.0;
        for (long base = 0; base < nstep; base += WS_RING) {
#pragma unroll
            for (int u = 0; u < WS_RING; ++u) {
                const long step = base + u;
                if (step >= 2 && step < nstep) { // consume sub-tile step-2 from stage (u+2)%4
                    if (step == 2) load_frags((u + 2) % WS_RING, 0, af, bf);
#pragma unroll
                    for (int ks = 0; ks < WS_ROWS / 4; ++ks) {
                        double an[NA], bn[NB];
                        if (ks + 1 < WS_ROWS / 4) load_frags((u + 2) % WS_RING, ks + 1, an, bn);
                        else                      load_frags((u + 3) % WS_RING, 0, an, bn); // next step's slot (unused garbage at the tail)
#pragma unroll
                        for (int i = 0; i < NA; ++i)
#pragma unroll
                            for (int j = 0; j < NB; ++j)
                                if ((4 * i + 3 < NT || wa + 4 * i < NT) && (2 * j + 1 < NT || wb + 2 * j < NT) && !(dbg & 2))
                                    acc[i][j] = mfma_f64(af[i], bf[j], acc[i][j]);
#pragma unroll
                        for (int i = 0; i < NA; ++i) af[i] = an[i];
#pragma unroll
                        for (int j = 0; j < NB; ++j) bf[j] = bn[j];
                    }
                }
                __syncthreads();
            }
        }
        if (!SYM) {
            double *slab = slabs + (size_t)blockIdx.x * nao * nao;
#pragma unroll
            for (int i = 0; i < NA; ++i)
#pragma unroll
                for (int j = 0; j < NB; ++j) {
                    if (wa + 4 * i >= NT || wb + 2 * j >= NT) continue;
                    const int b = 16 * (wb + 2 * j) + li;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int a = 16 * (wa + 4 * i) + lk + 4 * r;
                        if (a < nao && b < nao) slab[(size_t)a * nao + b] = acc[i][j][r];
                    }
                }
        } else {
            // the ring is dead after the last barrier of the loop: reuse it as M[a][b], ld NCOL+1
            constexpr int LDM = C::NCOL + 1;
            static_assert(LDM * C::NCOL <= 2 * WS_RING * TILE, "M tile must fit in the ring");
            double *M = ring;
#pragma unroll
            for (int i = 0; i < NA; ++i)
#pragma unroll
                for (int j = 0; j < NB; ++j) {
                    if (wa + 4 * i >= NT || wb + 2 * j >= NT) continue;
                    const int b = 16 * (wb + 2 * j) + li;
#pragma unroll
                    for (int r = 0; r < 4; ++r) M[(16 * (wa + 4 * i) + lk + 4 * r) * LDM + b] = acc[i][j][r];
                }
        }
    } else {
        // -------------------------------------------------------- loader role
        __builtin_amdgcn_s_setprio(3); // see k_vxc_ws
        const int lt = tid - 512, row = lt >> 5, seg = lt & 31;
        const double *c0 = coef, *c1 = coef + (size_t)ngrid, *c2 = coef + 2 * (size_t)ngrid,
                     *c3 = coef + 3 * (size_t)ngrid;
        double p0[2][2 * C::JN], p1[2][2 * C::JN], p2[2][2 * C::JN], p3[2][2 * C::JN]; // sub-tiles s and s+1 in flight
        double k0[2], k1[2], k2[2], k3[2];
        // unconditional issue, whole-plane descriptors + tile offset in an SGPR, drain steps out of range: see k_vxc_ws
        const long plane = ngrid * (long)nao;
        const unsigned voff = (unsigned)(row * nao + 2 * seg) * 8u, koff = (unsigned)row * 8u;
        const unsigned tile_b = (unsigned)(WS_ROWS * nao) * 8u, ktile_b = WS_ROWS * 8u;
        const unsigned plane_b = (unsigned)(plane * 8), coef_b = (unsigned)(ngrid * 8);
        const __amdgpu_buffer_rsrc_t r0 = plane_rsrc(ao, plane), r1 = plane_rsrc(GRAD ? gx : ao, plane),
                                     r2 = plane_rsrc(GRAD ? gy : ao, plane), r3 = plane_rsrc(GRAD ? gz : ao, plane),
                                     q0 = plane_rsrc(c0, ngrid), q1 = plane_rsrc(GRAD ? c1 : c0, ngrid),
                                     q2 = plane_rsrc(GRAD ? c2 : c0, ngrid), q3 = plane_rsrc(GRAD ? c3 : c0, ngrid);
        auto issue = [&](int set, unsigned s) {
            const bool live = s < (unsigned)nloc && !(dbg & 1);
            const unsigned t = ws_tile((unsigned)ntile, blockIdx.x, s, gridDim.x, rev);
            const unsigned so = live ? t * tile_b : plane_b, ko = live ? t * ktile_b : coef_b;
            k0[set] = buf_load_f64(q0, koff, ko);
            if (GRAD) {
                k1[set] = buf_load_f64(q1, koff, ko);
                k2[set] = buf_load_f64(q2, koff, ko);
                k3[set] = buf_load_f64(q3, koff, ko);
            }
            buf_load_row<C::JN, VEC, 512>(r0, voff, so, p0[set]);
            if (GRAD) {
                buf_load_row<C::JN, VEC, 512>(r1, voff, so, p1[set]);
                buf_load_row<C::JN, VEC, 512>(r2, voff, so, p2[set]);
                buf_load_row<C::JN, VEC, 512>(r3, voff, so, p3[set]);
            }
        };
        issue(0, 0);
        __builtin_amdgcn_sched_barrier(0); // keep program order: set 0 must be the OLDER one at the loop header (see k_rho_ws*)
        issue(1, 1);
        __builtin_amdgcn_sched_barrier(0);
        for (long base = 0; base < nstep; base += WS_RING) {
#pragma unroll
            for (int u = 0; u < WS_RING; ++u) {
                const long step = base + u;
                const int set = u & 1;
                if (!(dbg & 32)) {
                double *P = Ps + u * TILE, *Q = Qs + u * TILE;
#pragma unroll
                    for (int j = 0; j < C::JN; ++j) {
                        const int c = 64 * j + 2 * seg;
                        if (c < C::NCOL) {
                            const double a0 = p0[set][2 * j], b0 = p0[set][2 * j + 1];
                            double qa = k0[set] * a0, qb = k0[set] * b0;
                            if (GRAD && !(dbg & 64)) {
                                qa += k1[set] * p1[set][2 * j] + k2[set] * p2[set][2 * j] + k3[set] * p3[set][2 * j];
                                qb += k1[set] * p1[set][2 * j + 1] + k2[set] * p2[set][2 * j + 1] + k3[set] * p3[set][2 * j + 1];
                            } else if (GRAD) { // diagnostics: no fp64 FMAs, the gradient registers only kept alive
                                asm volatile("" ::"v"(p1[set][2 * j]), "v"(p2[set][2 * j]), "v"(p3[set][2 * j]), "v"(p1[set][2 * j + 1]), "v"(p2[set][2 * j + 1]), "v"(p3[set][2 * j + 1]));
                            }
                            if (!(dbg & 8)) {
                                *reinterpret_cast<double2 *>(&Q[row * C::LDX + c]) = make_double2(qa, qb);
                                *reinterpret_cast<double2 *>(&P[row * C::LDX + c]) = make_double2(a0, b0);
                            } else { // diagnostics: no LDS writes
                                asm volatile("" ::"v"(qa), "v"(qb), "v"(a0), "v"(b0));
                            }
                        }
                    }
                issue(set, (unsigned)step + 2u);
                }
                __syncthreads();
            }
        }
    }
    if (SYM) { // all sixteen waves: slab = M + M^T, coalesced stores
        constexpr int LDM = C::NCOL + 1;
        const double *M = ring;
        __syncthreads();
        double *slab = slabs + (size_t)blockIdx.x * nao * nao;
        for (int e = tid; e < nao * nao; e += W16_THREADS) {
            const int a = e / nao, b = e - a * nao;
            slab[e] = M[a * LDM + b] + M[b * LDM + a]; // (x + y) == (y + x): bitwise symmetric
        }
    }
}

// ------------------------------------------------------------------ rho ----
// rho_g = sum_v X[g][v] AO[g][v],  grad rho_g = 2 sum_v X[g][v] dAO[g][v],  X = AO . Ds.
// MFMA wave w owns column tile w: its 16 columns of Ds stay in 4*NT registers.
template <int NT, bool GRAD, bool VEC>
__global__ __launch_bounds__(W16_THREADS) void k_rho_ws16(long ngrid, int nao,
                                                         const double *__restrict__ ao,
                                                         const double *__restrict__ gx,
                                                         const double *__restrict__ gy,
                                                         const double *__restrict__ gz,
                                                         const double *__restrict__ dm,
                                                         double *__restrict__ rho,
                                                         double *__restrict__ grad,
                                                         double *__restrict__ sigma, int dbg)
{
    using C = W16Cfg<NT>;
    constexpr int NKS = 4 * NT; // k-steps over the padded AO index
    constexpr int ATILE = WS_ROWS * C::LDA, XTILE = WS_ROWS * C::LDX;
    __shared__ double As[WS_RING * ATILE];
    __shared__ double Xs[2 * XTILE];

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const long ntile = (ngrid + WS_ROWS - 1) / WS_ROWS;
    const long nloc = (ntile > (long)blockIdx.x) ? (ntile - blockIdx.x + gridDim.x - 1) / gridDim.x : 0;
    const long nstep = nloc + 2;
    const int rev = (dbg >> 16) & 1;

    if (wave < 8) {
        // ---------------------------------------------------------- MFMA role
        const int lane = tid & 63, li = lane & 15, lk = lane >> 4;
        const int nks = (nao + 3) >> 2; // k-steps that carry data
        const int tcol = min(wave, NT - 1);
        double dreg[NKS];
        { // Ds = (D + D^T)/2, zero outside nao x nao, straight from the caller's matrix
            const int n = 16 * tcol + li;
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
                const int k = 4 * ks + lk;
                const bool in = k < nao && n < nao;
                const int kc = in ? k : 0, nc = in ? n : 0;
                const double v = 0.5 * (dm[(size_t)kc * nao + nc] + dm[(size_t)nc * nao + kc]);
                dreg[ks] = in ? v : 0.0;
            }
        }
        for (long base = 0; base < nstep; base += WS_RING) {
#pragma unroll
            for (int u = 0; u < WS_RING; ++u) {
                const long step = base + u;
                if (step >= 1 && step <= nloc && wave < NT) { // X(step-1) from AO stage (u+3)%4
                    const double *ap = As + ((u + 3) % WS_RING) * ATILE + li * C::LDA + lk;
                    double *X = Xs + ((u + 1) & 1) * XTILE;
                    d4 acc = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
                    for (int ks = 0; ks < NKS; ++ks)
                        if ((ks < NKS - 3 || ks < nks) && !(dbg & 2)) acc = mfma_f64(ap[4 * ks], dreg[ks], acc); // same chain as k_rho_ws: X is bit-identical
#pragma unroll
                    for (int r = 0; r < 4; ++r) X[(lk + 4 * r) * C::LDX + 16 * wave + li] = acc[r];
                }
                __syncthreads();
            }
        }
    } else {
        // -------------------------------------------------------- loader role
        __builtin_amdgcn_s_setprio(3);
        const int lt = tid - 512, row = lt >> 5, seg = lt & 31;
        double ph[2][2 * C::JN];                                           // AO of sub-tiles s, s+1
        double pgx[2][2 * C::JN], pgy[2][2 * C::JN], pgz[2][2 * C::JN];   // gradients, consumed two steps later

        auto row_of = [&](long s) { return (long)ws_tile((unsigned)ntile, blockIdx.x, (unsigned)s, gridDim.x, rev) * WS_ROWS + row; };
        const long plane = ngrid * (long)nao;
        const unsigned voff = (unsigned)(row * nao + 2 * seg) * 8u;
        const unsigned tile_b = (unsigned)(WS_ROWS * nao) * 8u, plane_b = (unsigned)(plane * 8);
        const __amdgpu_buffer_rsrc_t r0 = plane_rsrc(ao, plane), r1 = plane_rsrc(GRAD ? gx : ao, plane),
                                     r2 = plane_rsrc(GRAD ? gy : ao, plane), r3 = plane_rsrc(GRAD ? gz : ao, plane);
        auto tile_off = [&](unsigned s) {
            return s < (unsigned)nloc && !(dbg & 1) ? ws_tile((unsigned)ntile, blockIdx.x, s, gridDim.x, rev) * tile_b : plane_b;
        };
        auto issue_ao = [&](int set, unsigned s) { buf_load_row<C::JN, VEC, 512>(r0, voff, tile_off(s), ph[set]); };
        auto issue_grad = [&](int set, unsigned s) {
            const unsigned so = tile_off(s);
            buf_load_row<C::JN, VEC, 512>(r1, voff, so, pgx[set]);
            buf_load_row<C::JN, VEC, 512>(r2, voff, so, pgy[set]);
            buf_load_row<C::JN, VEC, 512>(r3, voff, so, pgz[set]);
        };
        // Prologue in the loop's own issue order (AO set 0, gradients set 0, AO set 1, gradients set 1; the
        // gradient loads here are dead -- zero-record descriptors, no traffic -- and land in registers the
        // loop overwrites before it reads them), pinned with scheduling barriers: the wait the compiler puts
        // at the loop header is the MINIMUM over the entry path and the back edge of "loads younger than the
        // set consumed first", so a prologue that issues fewer or reordered loads turns the header wait into a
        // near-drain of the whole prefetch once per trip (seen in the ISA of round 1: vmcnt(0) at the header).
        issue_ao(0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (GRAD) issue_grad(0, (unsigned)nloc);
        __builtin_amdgcn_sched_barrier(0);
        issue_ao(1, 1);
        __builtin_amdgcn_sched_barrier(0);
        if (GRAD) issue_grad(1, (unsigned)nloc);
        __builtin_amdgcn_sched_barrier(0);

        for (long base = 0; base < nstep; base += WS_RING) {
#pragma unroll
            for (int u = 0; u < WS_RING; ++u) {
                const long step = base + u;
                const int set = u & 1;
                { // (a) AO(step) -> ring slot u, then refill the register set with AO(step+2)
                    double *A = As + u * ATILE;
#pragma unroll
                    for (int j = 0; j < C::JN; ++j) {
                        const int c = 64 * j + 2 * seg;
                        if (c < C::NCOL)
                            *reinterpret_cast<double2 *>(&A[row * C::LDA + c]) = make_double2(ph[set][2 * j], ph[set][2 * j + 1]);
                    }
                    issue_ao(set, (unsigned)step + 2u);
                }
                { // (b) row dots of sub-tile step-2 (see k_rho_ws); the row's 32 lanes are one half-wave
                    const bool in_range = step >= 2 && step - 2 < nloc;
                    const long g = row_of(in_range ? step - 2 : 0);
                    const bool row_ok = in_range && g < ngrid;
                    const double *A = As + ((u + 2) % WS_RING) * ATILE;
                    const double *X = Xs + (u & 1) * XTILE;
                    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
#pragma unroll
                    for (int j = 0; j < C::JN; ++j) {
                        const int c = 64 * j + 2 * seg;
                        if (c < C::NCOL) {
                            const double2 x = *reinterpret_cast<const double2 *>(&X[row * C::LDX + c]);
                            const double2 p = *reinterpret_cast<const double2 *>(&A[row * C::LDA + c]);
                            s0 += x.x * p.x + x.y * p.y;
                            if (GRAD) {
                                s1 += x.x * pgx[set][2 * j] + x.y * pgx[set][2 * j + 1];
                                s2 += x.x * pgy[set][2 * j] + x.y * pgy[set][2 * j + 1];
                                s3 += x.x * pgz[set][2 * j] + x.y * pgz[set][2 * j + 1];
                            }
                        }
                    }
                    s0 = half32_sum_hi(s0);
                    if (GRAD) {
                        s1 = half32_sum_hi(s1);
                        s2 = half32_sum_hi(s2);
                        s3 = half32_sum_hi(s3);
                    }
                    if (seg == 16 && row_ok) {
                        rho[g] = s0;
                        if (GRAD) {
                            const double ax = 2.0 * s1, ay = 2.0 * s2, az = 2.0 * s3;
                            grad[3 * g + 0] = ax;
                            grad[3 * g + 1] = ay;
                            grad[3 * g + 2] = az;
                            sigma[g] = ax * ax + ay * ay + az * az;
                        }
                    }
                }
                if (GRAD) issue_grad(set, (unsigned)step); // (c) gradients of sub-tile `step`, consumed at step+2
                __syncthreads();
            }
        }
    }
}

} // namespace qcdft
