// One-pass sweep for small bases (nao <= 32): density, pointwise functional and the Vxc contraction of a
// 16-point sub-tile without leaving the wave, so a call is this kernel plus the slab sum instead of
// rho -> xc_points -> vxc -> reduce.  At these sizes (H2O/def2-SVP: 24 functions, 34 k points) every kernel of
// the four-launch form lasts about as long as its dispatch and the call is a chain of launch latencies
// (profiles/r03_graph_time.txt: 30 us, 26 us replayed as a graph); the planes are also read once instead of twice.
//
// A wave owns sub-tiles wave * n_WG + workgroup (+ 8 n_WG per round).  Per sub-tile:
//   1. the four planes land in registers in the RESULT layout of v_mfma_f64_16x16x4 (lane (lk, li) holds rows
//      lk + 4r, r = 0..3, column 16t + li): rows of 128 contiguous bytes;
//   2. X = AO . Ds on the matrix pipe (A fragments AO[g0 + li][4ks + lk] are a second, L1-served read of the AO
//      plane: the contraction index has to sit on lk); X comes out in the same result layout;
//   3. rho, grad rho = row sums of X * plane: products in place, DPP row sums -- every lane of DPP row lk ends
//      with the four rows lk + 4r;
//   4. lanes li < 4 evaluate the functional at row lk + 4 li (src/dft_solver.cu:309-344, :382-432, :434-513 per-point
//      bodies, xc_functionals.hpp) and hand the coefficients back to their DPP row;
//   5. V += Q^T P with Q = sum_c coef_c plane_c: the registers of step 1 ARE the operand fragments, k-step r
//      contracting rows lk + 4r (the order of the contraction index is free as long as both operands agree).
// The eight waves' accumulators are added in a fixed order through LDS; one slab and one Exc partial per
// workgroup, summed by k_reduce_slabs8<false, true> like the slabs of the wave-specialised kernels: bitwise
// reproducible.  SYM (B3LYP, symmetrize_matrix_kernel :515-527): slab = M + M^T.
//
// Tried and dropped: warming loads of the next sub-tile ahead of the functional (one dword per 64 B of each plane, so
// that the end-of-iteration loads hit L2): 3-9 % slower at 150-300 k points (tools/tiny_time.py scan) -- the loop is
// not waiting for HBM, the extra requests only compete with the real ones.
//
// References replaced: src/dft_solver.cu:294-307, :346-380 (density), :309-513 (fused passes), :541-548 (Vxc GEMM),
// :285-292 (Exc sum), sequenced as in :559-672.
#pragma once
#include <hip/hip_runtime.h>
#include "device_util.hpp"
#include "xc_functionals.hpp"

namespace qcdft {

// tools/tiny_phase_probe.hip defines QCDFT_TINY_STAMPS: lane 0 of every wave leaves 100 MHz time stamps at the phase
// boundaries of its FIRST sub-tile (8 per wave).  Not compiled into libdft.so.
#ifdef QCDFT_TINY_STAMPS
__device__ unsigned long long *g_tiny_stamps;
#define TINY_STAMP(i, drain)                                                                                         \
    do {                                                                                                                \
        if (drain) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                                          \
        if (first_pass && lane == 0) g_tiny_stamps[((size_t)blockIdx.x * TN_WAVES + wave) * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); \
    } while (0)
#else
#define TINY_STAMP(i, drain) do { } while (0)
#endif

constexpr int TN_THREADS = 512;
constexpr int TN_WAVES = TN_THREADS / 64;
constexpr int TN_MAX_NAO = 32;

// TYPE 0 LDA, 1 GGA (PBE), 2 B3LYP
template <int NT, int TYPE, bool SYM>
__global__ __launch_bounds__(TN_THREADS) void k_sweep_tiny(long ngrid, int nao,
                                                          const double *__restrict__ ao,
                                                          const double *__restrict__ gx,
                                                          const double *__restrict__ gy,
                                                          const double *__restrict__ gz,
                                                          const double *__restrict__ dm,
                                                          const double *__restrict__ w,
                                                          double *__restrict__ slabs,
                                                          double *__restrict__ partial, int quirks)
{
    constexpr bool GRAD = TYPE != 0;
    constexpr int NCOL = 16 * NT, NKS = 4 * NT, LDM = NCOL + 1;
    __shared__ double Ms[4 * NCOL * LDM];
    __shared__ double es[TN_WAVES];
    __shared__ double Dl[NT * NKS * 64];

    const int tid = threadIdx.x, lane = tid & 63, li = lane & 15, lk = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const long ntile = (ngrid + 15) / 16;
    const long plane = ngrid * (long)nao;
    [[maybe_unused]] bool first_pass = true;
    TINY_STAMP(0, false);
    const __amdgpu_buffer_rsrc_t r0 = plane_rsrc(ao, plane), r1 = plane_rsrc(GRAD ? gx : ao, plane),
                                 r2 = plane_rsrc(GRAD ? gy : ao, plane), r3 = plane_rsrc(GRAD ? gz : ao, plane),
                                 rw = plane_rsrc(w, ngrid);

    // Per-lane byte offsets inside a sub-tile; the k-step / column-tile part is an immediate, the row group r an
    // SGPR.  Columns >= nao read the next row (or zeros past the plane) and are replaced by zeros after the load.
    const unsigned baseA = (unsigned)(li * nao + lk) * 8u, baseP = (unsigned)(lk * nao + li) * 8u;
    const unsigned rstep = (unsigned)(4 * nao) * 8u;
    const int rr = li & 3; // the row this lane evaluates when li < 4: lk + 4 rr
    const bool colok[2] = {li < nao, 16 + li < nao};

    d4 acc[NT][NT];
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = (d4){0.0, 0.0, 0.0, 0.0};
    double esum = 0.0;

    double p0[4][NT], p1[4][NT], p2[4][NT], p3[4][NT];
    auto load_planes = [&](unsigned soff) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                p0[r][t] = buf_load_f64(r0, baseP + 128u * t, soff + r * rstep);
                if (GRAD) {
                    p1[r][t] = buf_load_f64(r1, baseP + 128u * t, soff + r * rstep);
                    p2[r][t] = buf_load_f64(r2, baseP + 128u * t, soff + r * rstep);
                    p3[r][t] = buf_load_f64(r3, baseP + 128u * t, soff + r * rstep);
                }
            }
    };
    auto mask_planes = [&]() {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int t = 0; t < NT; ++t)
                if (!colok[t]) {
                    p0[r][t] = 0.0;
                    if (GRAD) p1[r][t] = p2[r][t] = p3[r][t] = 0.0;
                }
    };

    double af[NKS], wt = 0.0;
    auto issue_tile = [&](long tile) { // the loads of one sub-tile; masked at the top of the loop body, once they have landed
        const unsigned soff = (unsigned)(tile * 16 * nao) * 8u; // planes stay below 4 GiB (host check)
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) af[ks] = buf_load_f64(r0, baseA + 32u * ks, soff);
        load_planes(soff);
        wt = buf_load_f64(rw, (unsigned)(lk + 4 * rr) * 8u, (unsigned)(tile * 16) * 8u);
    };
    const long stride = (long)gridDim.x * TN_WAVES;
    // round s hands tile s * stride + wave * gridDim.x + blockIdx.x to this wave: a partial last round goes to wave 0 of
    // every workgroup, then wave 1 ... -- one extra sub-tile per CU, on a SIMD whose other wave has finished -- instead of
    // to all eight waves of the first few workgroups
    long tile = (long)wave * gridDim.x + blockIdx.x;
    if (tile < ntile) issue_tile(tile); // in flight while the density matrix is staged
    // Ds = (D + D^T)/2 as B fragments, zero outside nao x nao, straight from the caller's matrix; kept in LDS
    // (the same 16 NT^2 values per lane for every wave; in registers they cost 8 NT^2 VGPRs through the functional)
    if (tid < 64) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int n = 16 * t + li;
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
                const int k = 4 * ks + lk;
                const bool in = k < nao && n < nao;
                const int kc = in ? k : 0, nc = in ? n : 0;
                const double v = 0.5 * (dm[(size_t)kc * nao + nc] + dm[(size_t)nc * nao + kc]);
                Dl[(t * NKS + ks) * 64 + lane] = in ? v : 0.0;
            }
        }
    }
    __syncthreads();
    TINY_STAMP(1, false);

    for (; tile < ntile; tile += stride) {
        const long g = tile * 16 + lk + 4 * rr;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks)
            if (4 * ks + lk >= nao) af[ks] = 0.0;
        mask_planes();
        const double wt_now = wt;
        TINY_STAMP(2, true);

        // X = AO . Ds
        d4 x[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            x[t] = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks)
                if (4 * ks < nao) x[t] = mfma_f64(af[ks], Dl[(t * NKS + ks) * 64 + lane], x[t]);
        }
        // row sums: every lane of DPP row lk ends with rows lk + 4r and keeps the one it evaluates (lk + 4 rr)
        double rho = 0.0, d1 = 0.0, d2 = 0.0, d3 = 0.0;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                a0 += x[t][r] * p0[r][t];
                if (GRAD) {
                    a1 += x[t][r] * p1[r][t];
                    a2 += x[t][r] * p2[r][t];
                    a3 += x[t][r] * p3[r][t];
                }
            }
            a0 = row16_sum(a0);
            rho = rr == r ? a0 : rho;
            if (GRAD) {
                a1 = row16_sum(a1);
                a2 = row16_sum(a2);
                a3 = row16_sum(a3);
                d1 = rr == r ? a1 : d1;
                d2 = rr == r ? a2 : d2;
                d3 = rr == r ? a3 : d3;
            }
        }
        TINY_STAMP(3, true);
        // the functional at row lk + 4 rr, on lanes li < 4
        xc::PointXC p = {0.0, 0.0, 0.0, 0.0, 0.0};
        if (li < 4 && g < ngrid) {
            if (TYPE == 0) {
                p = xc::lda_point(rho, wt_now, quirks != 0);
            } else {
                const double ax = 2.0 * d1, ay = 2.0 * d2, az = 2.0 * d3;
                const double sg = ax * ax + ay * ay + az * az;
                if (TYPE == 1) p = xc::gga_point(rho, sg, ax, ay, az, wt_now, quirks != 0);
                else           p = xc::b3lyp_point(rho, sg, ax, ay, az, wt_now);
            }
            esum += wt_now * p.exc;
        }
        TINY_STAMP(4, true);
        // V += Q^T P, k-step r = rows lk + 4r
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int src = (lane & 48) | r;
            const double k0 = __shfl(p.c0, src, 64);
            double q[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) q[t] = k0 * p0[r][t];
            if (GRAD) {
                const double k1 = __shfl(p.c1, src, 64), k2 = __shfl(p.c2, src, 64), k3 = __shfl(p.c3, src, 64);
#pragma unroll
                for (int t = 0; t < NT; ++t) q[t] += k1 * p1[r][t] + k2 * p2[r][t] + k3 * p3[r][t];
            }
#pragma unroll
            for (int ta = 0; ta < NT; ++ta)
#pragma unroll
                for (int tb = 0; tb < NT; ++tb) acc[ta][tb] = mfma_f64(q[ta], p0[r][tb], acc[ta][tb]);
        }
        TINY_STAMP(5, true);
        first_pass = false;
        if (tile + stride < ntile) issue_tile(tile + stride);
    }

    // the eight waves' sums in a fixed order: waves 0-3 store, waves 4-7 add to the slot of wave - 4, then
    // ((M0 + M1) + (M2 + M3)) per element
    for (int m = 32; m >= 1; m >>= 1) esum += __shfl_down(esum, m, 64);
    if (lane == 0) es[wave] = esum;
    double *M = Ms + (wave & 3) * NCOL * LDM;
    if (wave < 4) {
#pragma unroll
        for (int ta = 0; ta < NT; ++ta)
#pragma unroll
            for (int tb = 0; tb < NT; ++tb)
#pragma unroll
                for (int r = 0; r < 4; ++r) M[(16 * ta + lk + 4 * r) * LDM + 16 * tb + li] = acc[ta][tb][r];
    }
    __syncthreads();
    if (wave >= 4) {
#pragma unroll
        for (int ta = 0; ta < NT; ++ta)
#pragma unroll
            for (int tb = 0; tb < NT; ++tb)
#pragma unroll
                for (int r = 0; r < 4; ++r) M[(16 * ta + lk + 4 * r) * LDM + 16 * tb + li] += acc[ta][tb][r];
    }
    __syncthreads();
    first_pass = true;
    TINY_STAMP(6, false);
    double *slab = slabs + (size_t)blockIdx.x * nao * nao;
    constexpr int SL = NCOL * LDM;
    for (int e = tid; e < nao * nao; e += TN_THREADS) {
        const int a = e / nao, b = e - a * nao;
        const int ab = a * LDM + b, ba = b * LDM + a;
        const double mab = (Ms[ab] + Ms[SL + ab]) + (Ms[2 * SL + ab] + Ms[3 * SL + ab]);
        if (SYM) {
            const double mba = (Ms[ba] + Ms[SL + ba]) + (Ms[2 * SL + ba] + Ms[3 * SL + ba]);
            slab[e] = mab + mba; // (x + y) == (y + x): bitwise symmetric
        } else {
            slab[e] = mab;
        }
    }
    TINY_STAMP(7, true);
    if (tid == 0) partial[blockIdx.x] = ((es[0] + es[1]) + (es[2] + es[3])) + ((es[4] + es[5]) + (es[6] + es[7]));
}

} // namespace qcdft
