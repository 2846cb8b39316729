// Contraction kernels for nao > 128 (Anthracene/def2-SVP 246, def2-TZVP 494, C33.../def2-SVP
// 1150): the sweep is fp64-MFMA-bound there (AI 60-140 flop/B), so these are classic LDS-tiled
// GEMM tiles -- 512 threads = 8 waves in a 2x4 grid, workgroup tile 128 x 256, wave tile
// 64 x 64 = 4x4 MFMA tiles (128 accumulator VGPRs), BK = 16, double-buffered LDS, next stage's
// global loads issued before the current stage's MFMAs (in-wave staging: ~30 non-fp64
// instructions per 64 MFMAs).  Plane tiles are read with range-checked buffer loads
// (xc_ws_kernels.hpp): rows past the grid arrive as zeros, columns past nao only ever meet
// exact zeros (zero-padded Ds) or discarded output tiles.
//
//   k_rho_big : X = AO . Ds tile by tile (128 x 128, wave tile 64 x 32); after each column block the X tile goes through
//               LDS in four 32-row slabs so the row dots (rho, grad rho) run in the coalesced
//               (row, seg) mapping; partial row sums stay in registers across column blocks.
//   k_vxc_big : V[a-block 128][b-block 256] += Q^T . AO over one grid chunk, Q formed in
//               registers from the four planes while staging.  Workgroups that share a grid
//               chunk are placed on one XCD (blockIdx % 8 groups) so the planes are fetched
//               from HBM once per chunk and re-read from that XCD's L2.
//
// References replaced: src/dft_solver.cu:294-307,346-380 (rho), :309-513 pass 2 + :541-548 (Vxc).
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>
#include "xc_ws_kernels.hpp"

namespace qcdft {

constexpr int BG_THREADS = 512;
constexpr int BG_BM = 128, BG_BN = 256, BG_BK = 16;
constexpr int BG_LDA = BG_BK + 2;   // 18: A-operand reads conflict-free
constexpr int BG_LDB = BG_BN + 16;  // 272 = 16 (mod 32)
constexpr int BG_LDQ = BG_BM + 16;  // 144

// ------------------------------------------------------------------ rho ----
// Workgroup tile 128 rows x 128 columns of X (wave tile 64 x 32 = 4x2 MFMA tiles): the row-dot
// epilogue needs ~100 VGPRs of its own, which a 4x4 wave tile (128 accumulator VGPRs) spills.
constexpr int RB_BN = 128, RB_LDB = RB_BN + 16; // 144
constexpr int RB_BK = 32, RB_LDA = RB_BK + 2;   // 34 = 2 (mod 32): 64 MFMAs per wave between barriers
template <bool GRAD, bool VEC>
__global__ __launch_bounds__(BG_THREADS, 2) void k_rho_big(long ngrid, int nao, int NP,
                                                           const double *__restrict__ ao,
                                                           const double *__restrict__ gx,
                                                           const double *__restrict__ gy,
                                                           const double *__restrict__ gz,
                                                           const double *__restrict__ Dp,
                                                           double *__restrict__ rho,
                                                           double *__restrict__ grad,
                                                           double *__restrict__ sigma)
{
    constexpr int ASZ = BG_BM * RB_LDA, BSZ = RB_BK * RB_LDB; // doubles per stage
    __shared__ double lds[2 * (ASZ + BSZ)];                   // 143,360 B; the X slab aliases it
    double *const As = lds, *const Bs = lds + 2 * ASZ;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lk = lane >> 4;
    const int wm = wave >> 2, wn = wave & 3; // rows wm*64 + 16i (i<4), columns wn*32 + 16j (j<2)
    const long g0 = (long)blockIdx.x * BG_BM;
    const long plane = ngrid * (long)nao;
    const int nkc = (NP + RB_BK - 1) / RB_BK;

    // staging maps
    const int a_row = tid >> 2, a_kq = tid & 3;   // A: 128 rows x 4 octets of k
    const int b_row = tid >> 4, b_cq = tid & 15;  // B: 32 k-rows x 16 octets of n
    const unsigned a_voff = (unsigned)(a_row * nao + 8 * a_kq) * 8u;
    const __amdgpu_buffer_rsrc_t ra = plane_tile_rsrc(ao, plane, g0 * nao);
    // epilogue map: 32 rows x 16 segs, columns 32q + 2seg + {0,1}, q < 4
    const int e_row = tid >> 4, e_seg = tid & 15;

    double s0[4] = {0, 0, 0, 0}, s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0}, s3[4] = {0, 0, 0, 0};

    for (int n0 = 0; n0 < NP; n0 += RB_BN) {
        d4 acc[4][2];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = (d4){0.0, 0.0, 0.0, 0.0};

        double2 rav[4], rb[4];
        auto fetch = [&](int kc) {
            const unsigned soff = (unsigned)(kc * RB_BK) * 8u;
#pragma unroll
            for (int q = 0; q < 4; ++q) rav[q] = buf_load_pair2<VEC>(ra, a_voff + 16 * q, soff);
            const int k = kc * RB_BK + b_row, n = n0 + 8 * b_cq;
            const bool ok = k < NP && n < NP; // NP is a multiple of 16: an octet is inside or outside
            const double2 *src = reinterpret_cast<const double2 *>(Dp + (size_t)(ok ? k : 0) * NP + (ok ? n : 0));
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const double2 v = src[q];
                rb[q] = ok ? v : make_double2(0.0, 0.0);
            }
        };
        auto stash = [&](int buf) {
            double *A = As + buf * ASZ + a_row * RB_LDA + 8 * a_kq;
            double *B = Bs + buf * BSZ + b_row * RB_LDB + 8 * b_cq;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                *reinterpret_cast<double2 *>(A + 2 * q) = rav[q];
                *reinterpret_cast<double2 *>(B + 2 * q) = rb[q];
            }
        };
        fetch(0);
        stash(0);
        __syncthreads();
        for (int kc = 0; kc < nkc; ++kc) {
            const int buf = kc & 1;
            fetch(kc + 1); // unconditional (see k_vxc_big): past the last k-chunk the operands are out of range / masked to zero and never used
            const double *A = As + buf * ASZ + (wm * 64 + li) * RB_LDA + lk;
            const double *B = Bs + buf * BSZ + lk * RB_LDB + wn * 32 + li;
#pragma unroll
            for (int ks = 0; ks < RB_BK / 4; ++ks) {
                double af[4], bf[2];
#pragma unroll
                for (int i = 0; i < 4; ++i) af[i] = A[16 * i * RB_LDA + 4 * ks];
#pragma unroll
                for (int j = 0; j < 2; ++j) bf[j] = B[4 * ks * RB_LDB + 16 * j];
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = mfma_f64(af[i], bf[j], acc[i][j]);
            }
            stash(buf ^ 1);
            __syncthreads();
        }

        // Row dots of this 128 x 128 block of X.  The whole X tile goes to LDS at once (the staging
        // buffers are dead here), then four passes of 32 rows in the coalesced (row, seg) mapping with
        // the plane loads of pass p+1 in flight under the arithmetic of pass p -- the accumulators'
        // registers are free by then.  (Slab by slab with two barriers and an exposed L2 round trip
        // each, this epilogue was 31 % of the kernel at nao 246 and 16 % at 494.)
        {
            constexpr int XLD = RB_BN + 8; // 136: 128 x 136 doubles = 139,264 B <= the staging LDS
            double *Xs = lds;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        Xs[(wm * 64 + 16 * i + lk + 4 * r) * XLD + wn * 32 + 16 * j + li] = acc[i][j][r];
            double2 v[2][4][4]; // [set][plane][column group]
            const unsigned voff = (unsigned)(e_row * nao + n0 + 2 * e_seg) * 8u;
            auto issue = [&](auto S, int p) {
                constexpr int st = decltype(S)::value;
                const long row0 = g0 + 32 * p;                     // wave-uniform
                const bool live = row0 < ngrid;
                const long e0 = (live ? row0 : 0) * (long)nao;
                const __amdgpu_buffer_rsrc_t r0 = plane_tile_rsrc(ao, plane, e0, live);
#pragma unroll
                for (int q = 0; q < 4; ++q) v[st][0][q] = buf_load_pair2<VEC>(r0, voff, (unsigned)(32 * q) * 8u);
                if (GRAD) {
                    const __amdgpu_buffer_rsrc_t r1 = plane_tile_rsrc(gx, plane, e0, live);
                    const __amdgpu_buffer_rsrc_t r2 = plane_tile_rsrc(gy, plane, e0, live);
                    const __amdgpu_buffer_rsrc_t r3 = plane_tile_rsrc(gz, plane, e0, live);
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        v[st][1][q] = buf_load_pair2<VEC>(r1, voff, (unsigned)(32 * q) * 8u);
                        v[st][2][q] = buf_load_pair2<VEC>(r2, voff, (unsigned)(32 * q) * 8u);
                        v[st][3][q] = buf_load_pair2<VEC>(r3, voff, (unsigned)(32 * q) * 8u);
                    }
                }
            };
            auto dots = [&](auto S, int p) {
                constexpr int st = decltype(S)::value;
                double t0 = 0, t1 = 0, t2 = 0, t3 = 0;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const double2 x = *reinterpret_cast<const double2 *>(&Xs[(32 * p + e_row) * XLD + 32 * q + 2 * e_seg]);
                    t0 += x.x * v[st][0][q].x + x.y * v[st][0][q].y;
                    if (GRAD) {
                        t1 += x.x * v[st][1][q].x + x.y * v[st][1][q].y;
                        t2 += x.x * v[st][2][q].x + x.y * v[st][2][q].y;
                        t3 += x.x * v[st][3][q].x + x.y * v[st][3][q].y;
                    }
                }
                s0[p] += t0; s1[p] += t1; s2[p] += t2; s3[p] += t3; // rows past the grid loaded zeros
            };
            using E0 = std::integral_constant<int, 0>;
            using E1 = std::integral_constant<int, 1>;
            issue(E0{}, 0);
            __syncthreads();
            issue(E1{}, 1);
            dots(E0{}, 0);
            issue(E0{}, 2);
            dots(E1{}, 1);
            issue(E1{}, 3);
            dots(E0{}, 2);
            dots(E1{}, 3);
            __syncthreads();
        }
    }

#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const double t0 = row16_sum(s0[p]);
        double t1 = 0, t2 = 0, t3 = 0;
        if (GRAD) {
            t1 = row16_sum(s1[p]);
            t2 = row16_sum(s2[p]);
            t3 = row16_sum(s3[p]);
        }
        const long g = g0 + 32 * p + e_row;
        if (e_seg == 0 && g < ngrid) {
            rho[g] = t0;
            if (GRAD) {
                const double ax = 2.0 * t1, ay = 2.0 * t2, az = 2.0 * t3;
                grad[3 * g + 0] = ax;
                grad[3 * g + 1] = ay;
                grad[3 * g + 2] = az;
                sigma[g] = ax * ax + ay * ay + az * az;
            }
        }
    }
}

// 64-row variant: 256 threads = 4 waves (2 x 2, wave tile 32 x 64), BK = 16, 69.6 KB of LDS, so TWO
// workgroups share a CU and one's epilogue -- which is bound by the per-CU fill rate: 524 KB of plane
// tiles per 128-column block, ~22 us at the ~10 B/cycle a CU gets from HBM, against ~54 us of MFMA --
// runs under the other's MFMA loop.  (With one 128-row workgroup per CU that traffic was exposed:
// 31 % of the kernel at nao 246, 16 % at 494; fewer barriers and deeper prefetch inside the epilogue
// changed nothing.)
constexpr int R6_BM = 64, R6_BK = 16, R6_LDA = R6_BK + 2, R6_XLD = RB_BN + 8; // A ld 18, X ld 136
template <bool GRAD, bool VEC>
__global__ __launch_bounds__(256, 2) void k_rho_big64(long ngrid, int nao, int NP,
                                                      const double *__restrict__ ao,
                                                      const double *__restrict__ gx,
                                                      const double *__restrict__ gy,
                                                      const double *__restrict__ gz,
                                                      const double *__restrict__ Dp,
                                                      double *__restrict__ rho,
                                                      double *__restrict__ grad,
                                                      double *__restrict__ sigma)
{
    constexpr int ASZ = R6_BM * R6_LDA, BSZ = R6_BK * RB_LDB;    // doubles per stage
    constexpr int STAGE = 2 * (ASZ + BSZ), XSZ = R6_BM * R6_XLD; // 6912 / 8704 doubles
    __shared__ double lds[XSZ > STAGE ? XSZ : STAGE];            // 69,632 B; the X tile aliases the staging
    double *const As = lds, *const Bs = lds + 2 * ASZ;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lk = lane >> 4;
    const int wm = wave >> 1, wn = wave & 1; // rows wm*32 + 16i (i<2), columns wn*64 + 16j (j<4)
    const long g0 = (long)blockIdx.x * R6_BM;
    const long plane = ngrid * (long)nao;
    const int nkc = (NP + R6_BK - 1) / R6_BK;

    const int a_row = tid >> 2, a_kq = tid & 3;   // A: 64 rows x 4 quads of k
    const int b_row = tid >> 4, b_cq = tid & 15;  // B: 16 k-rows x 16 octets of n
    const unsigned a_voff = (unsigned)(a_row * nao + 4 * a_kq) * 8u;
    const __amdgpu_buffer_rsrc_t ra = plane_tile_rsrc(ao, plane, g0 * nao);
    const int e_row = tid >> 4, e_seg = tid & 15; // epilogue: 16 rows x 16 segs per pass

    double s0[4] = {0, 0, 0, 0}, s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0}, s3[4] = {0, 0, 0, 0};

    for (int n0 = 0; n0 < NP; n0 += RB_BN) {
        d4 acc[2][4];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = (d4){0.0, 0.0, 0.0, 0.0};

        double2 rav[2], rb[4];
        auto fetch = [&](int kc) {
            const unsigned soff = (unsigned)(kc * R6_BK) * 8u;
#pragma unroll
            for (int q = 0; q < 2; ++q) rav[q] = buf_load_pair2<VEC>(ra, a_voff + 16 * q, soff);
            const int k = kc * R6_BK + b_row, n = n0 + 8 * b_cq;
            const bool ok = k < NP && n < NP; // NP is a multiple of 16: an octet is inside or outside
            const double2 *src = reinterpret_cast<const double2 *>(Dp + (size_t)(ok ? k : 0) * NP + (ok ? n : 0));
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const double2 v = src[q];
                rb[q] = ok ? v : make_double2(0.0, 0.0);
            }
        };
        auto stash = [&](int buf) {
            double *A = As + buf * ASZ + a_row * R6_LDA + 4 * a_kq;
            double *B = Bs + buf * BSZ + b_row * RB_LDB + 8 * b_cq;
#pragma unroll
            for (int q = 0; q < 2; ++q) *reinterpret_cast<double2 *>(A + 2 * q) = rav[q];
#pragma unroll
            for (int q = 0; q < 4; ++q) *reinterpret_cast<double2 *>(B + 2 * q) = rb[q];
        };
        fetch(0);
        stash(0);
        __syncthreads();
        for (int kc = 0; kc < nkc; ++kc) {
            const int buf = kc & 1;
            fetch(kc + 1); // unconditional (see k_vxc_big)
            const double *A = As + buf * ASZ + (wm * 32 + li) * R6_LDA + lk;
            const double *B = Bs + buf * BSZ + lk * RB_LDB + wn * 64 + li;
#pragma unroll
            for (int ks = 0; ks < R6_BK / 4; ++ks) {
                double af[2], bf[4];
#pragma unroll
                for (int i = 0; i < 2; ++i) af[i] = A[16 * i * R6_LDA + 4 * ks];
#pragma unroll
                for (int j = 0; j < 4; ++j) bf[j] = B[4 * ks * RB_LDB + 16 * j];
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][j] = mfma_f64(af[i], bf[j], acc[i][j]);
            }
            stash(buf ^ 1);
            __syncthreads();
        }

        // row dots of this 64 x 128 block of X: whole tile to LDS, four passes of 16 rows, the plane loads
        // of pass p+1 in flight under the arithmetic of pass p
        {
            double *Xs = lds;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        Xs[(wm * 32 + 16 * i + lk + 4 * r) * R6_XLD + wn * 64 + 16 * j + li] = acc[i][j][r];
            double2 v[2][4][4]; // [set][plane][column group]
            const unsigned voff = (unsigned)(e_row * nao + n0 + 2 * e_seg) * 8u;
            auto issue = [&](auto S, int p) {
                constexpr int st = decltype(S)::value;
                const long row0 = g0 + 16 * p;                     // wave-uniform
                const bool live = row0 < ngrid;
                const long e0 = (live ? row0 : 0) * (long)nao;
                const __amdgpu_buffer_rsrc_t r0 = plane_tile_rsrc(ao, plane, e0, live);
#pragma unroll
                for (int q = 0; q < 4; ++q) v[st][0][q] = buf_load_pair2<VEC>(r0, voff, (unsigned)(32 * q) * 8u);
                if (GRAD) {
                    const __amdgpu_buffer_rsrc_t r1 = plane_tile_rsrc(gx, plane, e0, live);
                    const __amdgpu_buffer_rsrc_t r2 = plane_tile_rsrc(gy, plane, e0, live);
                    const __amdgpu_buffer_rsrc_t r3 = plane_tile_rsrc(gz, plane, e0, live);
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        v[st][1][q] = buf_load_pair2<VEC>(r1, voff, (unsigned)(32 * q) * 8u);
                        v[st][2][q] = buf_load_pair2<VEC>(r2, voff, (unsigned)(32 * q) * 8u);
                        v[st][3][q] = buf_load_pair2<VEC>(r3, voff, (unsigned)(32 * q) * 8u);
                    }
                }
            };
            auto dots = [&](auto S, int p) {
                constexpr int st = decltype(S)::value;
                double t0 = 0, t1 = 0, t2 = 0, t3 = 0;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const double2 x = *reinterpret_cast<const double2 *>(&Xs[(16 * p + e_row) * R6_XLD + 32 * q + 2 * e_seg]);
                    t0 += x.x * v[st][0][q].x + x.y * v[st][0][q].y;
                    if (GRAD) {
                        t1 += x.x * v[st][1][q].x + x.y * v[st][1][q].y;
                        t2 += x.x * v[st][2][q].x + x.y * v[st][2][q].y;
                        t3 += x.x * v[st][3][q].x + x.y * v[st][3][q].y;
                    }
                }
                s0[p] += t0; s1[p] += t1; s2[p] += t2; s3[p] += t3; // rows past the grid loaded zeros
            };
            using E0 = std::integral_constant<int, 0>;
            using E1 = std::integral_constant<int, 1>;
            issue(E0{}, 0);
            __syncthreads();
            issue(E1{}, 1);
            dots(E0{}, 0);
            issue(E0{}, 2);
            dots(E1{}, 1);
            issue(E1{}, 3);
            dots(E0{}, 2);
            dots(E1{}, 3);
            __syncthreads();
        }
    }

#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const double t0 = row16_sum(s0[p]);
        double t1 = 0, t2 = 0, t3 = 0;
        if (GRAD) {
            t1 = row16_sum(s1[p]);
            t2 = row16_sum(s2[p]);
            t3 = row16_sum(s3[p]);
        }
        const long g = g0 + 16 * p + e_row;
        if (e_seg == 0 && g < ngrid) {
            rho[g] = t0;
            if (GRAD) {
                const double ax = 2.0 * t1, ay = 2.0 * t2, az = 2.0 * t3;
                grad[3 * g + 0] = ax;
                grad[3 * g + 1] = ay;
                grad[3 * g + 2] = az;
                sigma[g] = ax * ax + ay * ay + az * az;
            }
        }
    }
}

// ------------------------------------------------------------------ Vxc ----
// blockIdx.x -> (xcd = b % 8, slot = b / 8); pair = slot % npair; chunk = xcd + 8 * (slot / npair).
// slabs: one nao x nao matrix per chunk; every (a-block, b-block) pair writes its own region.
template <bool GRAD, bool VEC>
__global__ __launch_bounds__(BG_THREADS, 2) void k_vxc_big(long ngrid, int nao, long chunk, int nB,
                                                           int npair,
                                                           const double *__restrict__ ao,
                                                           const double *__restrict__ gx,
                                                           const double *__restrict__ gy,
                                                           const double *__restrict__ gz,
                                                           const double *__restrict__ coef,
                                                           double *__restrict__ slabs)
{
    constexpr int QSZ = BG_BK * BG_LDQ, PSZ = BG_BK * BG_LDB;
    __shared__ double lds[2 * (QSZ + PSZ)]; // 106,496 B
    double *const Qs = lds, *const Ps = lds + 2 * QSZ;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lk = lane >> 4;
    const int wm = wave >> 2, wn = wave & 3;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int pair = slot % npair, ck = xcd + 8 * (slot / npair);
    const int a0 = (pair / nB) * BG_BM, b0 = (pair % nB) * BG_BN;
    const long glo = (long)ck * chunk, ghi = min(ngrid, glo + chunk);

    d4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (d4){0.0, 0.0, 0.0, 0.0};

    if (glo < ghi) {
        const int nst = (int)((ghi - glo + BG_BK - 1) / BG_BK);
        // staging maps: 16 rows x 32 column groups
        const int s_row = tid >> 5, s_cq = tid & 31;
        const unsigned q_voff = (unsigned)(s_row * nao + a0 + 4 * s_cq) * 8u; // 4 Q columns
        const unsigned p_voff = (unsigned)(s_row * nao + b0 + 8 * s_cq) * 8u; // 8 P columns
        const unsigned k_voff = (unsigned)s_row * 8u;
        const double *c0 = coef, *c1 = coef + (size_t)ngrid, *c2 = coef + 2 * (size_t)ngrid,
                     *c3 = coef + 3 * (size_t)ngrid;

        double2 q0[2], q1[2], q2[2], q3[2], pp[4];
        double k0 = 0, k1 = 0, k2 = 0, k3 = 0;
        // ONE descriptor per plane and per coefficient row for the whole chunk (rows >= ghi must not contribute: the
        // range ends at ghi), the stage selected by the SGPR offset of the loads; a stage past the end passes the
        // chunk size -- every lane out of range, no traffic, the loads still count in vmcnt -- so the fetch is issued
        // UNCONDITIONALLY.  (With `if (st + 1 < nst) fetch(st + 1)` the loaded registers reached the loop header
        // through a merge, the compiler copied them there and waited for 12 of the 16 loads of the NEXT stage before
        // this stage's first MFMA: the prefetch was mostly serialised, 65 % MFMA-busy, profiles/r03_pmc_sweep.json.)
        // The host keeps a chunk of a plane below 4 GiB.
        const long crow = ghi - glo;
        const __amdgpu_buffer_rsrc_t r0 = plane_rsrc(ao + glo * nao, crow * nao);
        const __amdgpu_buffer_rsrc_t r1 = plane_rsrc((GRAD ? gx : ao) + glo * nao, crow * nao);
        const __amdgpu_buffer_rsrc_t r2 = plane_rsrc((GRAD ? gy : ao) + glo * nao, crow * nao);
        const __amdgpu_buffer_rsrc_t r3 = plane_rsrc((GRAD ? gz : ao) + glo * nao, crow * nao);
        const __amdgpu_buffer_rsrc_t d0 = plane_rsrc(c0 + glo, crow), d1 = plane_rsrc((GRAD ? c1 : c0) + glo, crow),
                                     d2 = plane_rsrc((GRAD ? c2 : c0) + glo, crow), d3 = plane_rsrc((GRAD ? c3 : c0) + glo, crow);
        const unsigned p_end = (unsigned)(crow * nao * 8), k_end = (unsigned)(crow * 8);
        const unsigned p_step = (unsigned)(BG_BK * nao) * 8u, k_step = BG_BK * 8u;
        auto fetch = [&](int st) {
            const bool live = st < nst;
            const unsigned so = live ? (unsigned)st * p_step : p_end, ko = live ? (unsigned)st * k_step : k_end;
            k0 = buf_load_d1(d0, k_voff, ko);
#pragma unroll
            for (int h = 0; h < 2; ++h) q0[h] = buf_load_pair2<VEC>(r0, q_voff + 16 * h, so);
#pragma unroll
            for (int h = 0; h < 4; ++h) pp[h] = buf_load_pair2<VEC>(r0, p_voff + 16 * h, so);
            if (GRAD) {
                k1 = buf_load_d1(d1, k_voff, ko);
                k2 = buf_load_d1(d2, k_voff, ko);
                k3 = buf_load_d1(d3, k_voff, ko);
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    q1[h] = buf_load_pair2<VEC>(r1, q_voff + 16 * h, so);
                    q2[h] = buf_load_pair2<VEC>(r2, q_voff + 16 * h, so);
                    q3[h] = buf_load_pair2<VEC>(r3, q_voff + 16 * h, so);
                }
            }
        };
        auto stash = [&](int buf) {
            double *Q = Qs + buf * QSZ + s_row * BG_LDQ + 4 * s_cq;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                double qa = k0 * q0[h].x, qb = k0 * q0[h].y;
                if (GRAD) {
                    qa += k1 * q1[h].x + k2 * q2[h].x + k3 * q3[h].x;
                    qb += k1 * q1[h].y + k2 * q2[h].y + k3 * q3[h].y;
                }
                *reinterpret_cast<double2 *>(Q + 2 * h) = make_double2(qa, qb);
            }
            double *P = Ps + buf * PSZ + s_row * BG_LDB + 8 * s_cq;
#pragma unroll
            for (int h = 0; h < 4; ++h) *reinterpret_cast<double2 *>(P + 2 * h) = pp[h];
        };
        fetch(0);
        stash(0);
        __syncthreads();
        for (int st = 0; st < nst; ++st) {
            const int buf = st & 1;
            fetch(st + 1);
            const double *Q = Qs + buf * QSZ + lk * BG_LDQ + wm * 64 + li;
            const double *P = Ps + buf * PSZ + lk * BG_LDB + wn * 64 + li;
#pragma unroll
            for (int ks = 0; ks < BG_BK / 4; ++ks) {
                double af[4], bf[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    af[i] = Q[4 * ks * BG_LDQ + 16 * i];
                    bf[i] = P[4 * ks * BG_LDB + 16 * i];
                }
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][j] = mfma_f64(af[i], bf[j], acc[i][j]);
            }
            stash(buf ^ 1); // behind the last stage: zeros into the idle buffer
            __syncthreads();
        }
    }

    double *slab = slabs + (size_t)ck * nao * nao;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int b = b0 + wn * 64 + 16 * j + li;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int a = a0 + wm * 64 + 16 * i + lk + 4 * r;
                if (a < nao && b < nao) slab[(size_t)a * nao + b] = acc[i][j][r];
            }
        }
}

} // namespace qcdft
