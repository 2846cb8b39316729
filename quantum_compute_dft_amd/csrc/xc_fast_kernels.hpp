// Persistent, software-pipelined contraction kernels for nao <= 128 (NT = ceil(nao/16) <= 8
// MFMA column tiles): the sizes of every BASELINE config that fits dense ERIs
// (H2O/def2-SVP 24, Benzene/def2-SVP 114, sto-3g 7..80).
//
// Same mathematics as xc_kernels.hpp (k_rho_mfma / k_vxc_mfma, which stay the
// generic path for nao > 128); what changes is the data movement:
//   * one 512-thread workgroup per CU, persistent over 32-point sub-tiles
//     (sub-tile i of workgroup b is b + i*gridDim.x: neighbouring CUs stream
//     neighbouring HBM pages);
//   * every AO plane is read with 16-byte loads, 16 lanes covering 256
//     contiguous bytes of one grid row (thread = (row, seg), columns
//     32j + 2seg + {0,1}); loads for sub-tile t+1 are in flight while t computes;
//   * rho kernel: the symmetrised density matrix is STATIONARY in registers
//     (wave w owns column tile w: 4*NT doubles per lane), AO tile in LDS with
//     leading dimension = 2 (mod 32) doubles so the 16-row x 2-k MFMA A-operand
//     read touches 32 distinct bank pairs; X goes back through LDS (ld = 16 mod 32)
//     so the row-dot epilogue runs in the coalesced (row, seg) mapping and needs
//     only a 16-lane butterfly;
//   * vxc kernel: B[g,a] = sum_c coef_c[g] plane_c[g,a] is formed in registers
//     from the just-loaded planes and staged with AO in double-buffered LDS
//     (one barrier per sub-tile); 8 waves own a 4x2 grid of 2x4 MFMA tiles of
//     the NTxNT output, accumulated in registers across ALL sub-tiles of the
//     workgroup and written once as a slab (deterministic fixed-order reduce).
//
// References replaced: src/dft_solver.cu:294-307,346-380 (rho kernels),
// :309-513 pass-2 B rows + :541-548 cublasDgemm (Vxc).
#pragma once
#include <hip/hip_runtime.h>
#include "xc_kernels.hpp"

namespace qcdft {

constexpr int FK_ROWS = 32;     // grid points per sub-tile
constexpr int FK_THREADS = 512; // 8 waves

template <int NT> struct FastCfg {
    static constexpr int NCOL = 16 * NT;                      // padded AO columns
    static constexpr int JN = (NT + 1) / 2;                   // 32-column groups per row
    static constexpr int LDA = NCOL + 2;                      // = 2 or 18 (mod 32)
    static constexpr int LDX = ((NCOL + 31) / 32) * 32 + 16;  // = 16 (mod 32)
};

// 2 consecutive doubles of one plane row; `vec` (nao even) allows the 16-byte form.
__device__ __forceinline__ void load_pair(const double *__restrict__ rowp, int col, int nao,
                                          bool row_ok, bool vec, double &a, double &b)
{
    a = 0.0;
    b = 0.0;
    if (row_ok && col < nao) {
        if (vec) { // col even and nao even => col+1 < nao
            const double2 v = *reinterpret_cast<const double2 *>(rowp + col);
            a = v.x;
            b = v.y;
        } else {
            a = rowp[col];
            if (col + 1 < nao) b = rowp[col + 1];
        }
    }
}

// ------------------------------------------------------------------ rho ----
template <int NT, bool GRAD>
__global__ __launch_bounds__(FK_THREADS, 2) void k_rho_fast(long ngrid, int nao, int vec16,
                                                            const double *__restrict__ ao,
                                                            const double *__restrict__ gx,
                                                            const double *__restrict__ gy,
                                                            const double *__restrict__ gz,
                                                            const double *__restrict__ Dp,
                                                            double *__restrict__ rho,
                                                            double *__restrict__ grad,
                                                            double *__restrict__ sigma)
{
    using C = FastCfg<NT>;
    constexpr int NKS = 4 * NT; // k-steps over the padded AO index
    __shared__ double As[FK_ROWS * C::LDA];
    __shared__ double Xs[FK_ROWS * C::LDX];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lk = lane >> 4;
    const int row = tid >> 4, seg = tid & 15;
    const bool vec = vec16 != 0; // nao even and all plane pointers 16-byte aligned (host-checked)
    const int nks = (nao + 3) >> 2; // k-steps that carry data
    const long ntile = (ngrid + FK_ROWS - 1) / FK_ROWS;

    // stationary operand: column tile `wave` of the symmetrised, zero-padded density matrix
    double dreg[NKS];
    if (wave < NT) {
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks)
            dreg[ks] = Dp[(size_t)(4 * ks + lk) * C::NCOL + 16 * wave + li];
    }

    double phn[2 * C::JN]; // AO values of the NEXT sub-tile (prefetch)
    long t = blockIdx.x;
    if (t < ntile) {
        const long g = t * FK_ROWS + row;
        const double *rp = ao + (size_t)g * nao;
#pragma unroll
        for (int j = 0; j < C::JN; ++j)
            load_pair(rp, 32 * j + 2 * seg, nao, g < ngrid, vec, phn[2 * j], phn[2 * j + 1]);
    }

    for (; t < ntile; t += gridDim.x) {
        const long g = t * FK_ROWS + row;
        const bool row_ok = g < ngrid;
        // AO tile -> LDS (A operand); the values also stay in registers for the row dots
        double phc[2 * C::JN];
#pragma unroll
        for (int j = 0; j < C::JN; ++j) {
            const int c = 32 * j + 2 * seg;
            phc[2 * j] = phn[2 * j];
            phc[2 * j + 1] = phn[2 * j + 1];
            if (c < C::NCOL)
                *reinterpret_cast<double2 *>(&As[row * C::LDA + c]) = make_double2(phc[2 * j], phc[2 * j + 1]);
        }
        // gradient planes of THIS sub-tile and AO of the NEXT one fly during the MFMA phase
        double pgx[2 * C::JN], pgy[2 * C::JN], pgz[2 * C::JN];
        if (GRAD) {
            const size_t ro = (size_t)g * nao;
#pragma unroll
            for (int j = 0; j < C::JN; ++j) {
                const int c = 32 * j + 2 * seg;
                load_pair(gx + ro, c, nao, row_ok, vec, pgx[2 * j], pgx[2 * j + 1]);
                load_pair(gy + ro, c, nao, row_ok, vec, pgy[2 * j], pgy[2 * j + 1]);
                load_pair(gz + ro, c, nao, row_ok, vec, pgz[2 * j], pgz[2 * j + 1]);
            }
        }
        {
            const long tn = t + gridDim.x;
            const long gn = tn * FK_ROWS + row;
            const bool ok = tn < ntile && gn < ngrid;
            const double *rp = ao + (size_t)(ok ? gn : 0) * nao;
#pragma unroll
            for (int j = 0; j < C::JN; ++j)
                load_pair(rp, 32 * j + 2 * seg, nao, ok, vec, phn[2 * j], phn[2 * j + 1]);
        }
        __syncthreads();

        // X[:, tile wave] = AO_tile . Ds[:, tile wave]
        if (wave < NT) {
            d4 acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
            const double *a0p = &As[li * C::LDA + lk];
            const double *a1p = &As[(16 + li) * C::LDA + lk];
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
                if (ks < NKS - 3 || ks < nks) {
                    acc0 = mfma_f64(a0p[4 * ks], dreg[ks], acc0);
                    acc1 = mfma_f64(a1p[4 * ks], dreg[ks], acc1);
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                Xs[(lk + 4 * r) * C::LDX + 16 * wave + li] = acc0[r];
                Xs[(16 + lk + 4 * r) * C::LDX + 16 * wave + li] = acc1[r];
            }
        }
        __syncthreads();

        // row dots in the (row, seg) mapping
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
#pragma unroll
        for (int j = 0; j < C::JN; ++j) {
            const int c = 32 * j + 2 * seg;
            if (c < C::NCOL) {
                const double2 x = *reinterpret_cast<const double2 *>(&Xs[row * C::LDX + c]);
                s0 += x.x * phc[2 * j] + x.y * phc[2 * j + 1];
                if (GRAD) {
                    s1 += x.x * pgx[2 * j] + x.y * pgx[2 * j + 1];
                    s2 += x.x * pgy[2 * j] + x.y * pgy[2 * j + 1];
                    s3 += x.x * pgz[2 * j] + x.y * pgz[2 * j + 1];
                }
            }
        }
#pragma unroll
        for (int m = 1; m < 16; m <<= 1) {
            s0 += __shfl_xor(s0, m, 64);
            if (GRAD) {
                s1 += __shfl_xor(s1, m, 64);
                s2 += __shfl_xor(s2, m, 64);
                s3 += __shfl_xor(s3, m, 64);
            }
        }
        if (seg == 0 && row_ok) {
            rho[g] = s0;
            if (GRAD) {
                const double ax = 2.0 * s1, ay = 2.0 * s2, az = 2.0 * s3;
                grad[3 * g + 0] = ax;
                grad[3 * g + 1] = ay;
                grad[3 * g + 2] = az;
                sigma[g] = ax * ax + ay * ay + az * az;
            }
        }
        // Two barriers per sub-tile suffice: As is rewritten after barrier 2 (every wave has left
        // the MFMA phase, the only reader of As); Xs is rewritten after the NEXT barrier 1, which
        // every wave reaches only after finishing these Xs reads.
    }
}

// ------------------------------------------------------------------ Vxc ----
template <int NT, bool GRAD>
__global__ __launch_bounds__(FK_THREADS, 2) void k_vxc_fast(long ngrid, int nao, int vec16,
                                                            const double *__restrict__ ao,
                                                            const double *__restrict__ gx,
                                                            const double *__restrict__ gy,
                                                            const double *__restrict__ gz,
                                                            const double *__restrict__ coef,
                                                            double *__restrict__ slabs)
{
    using C = FastCfg<NT>;
    constexpr int TILE = FK_ROWS * C::LDX;
    __shared__ double Ps[2 * TILE];
    __shared__ double Qs[2 * TILE];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lk = lane >> 4;
    const int row = tid >> 4, seg = tid & 15;
    const int wa = wave >> 1, wb = wave & 1; // 4 x 2 wave grid over the NT x NT tiles
    const bool vec = vec16 != 0; // nao even and all plane pointers 16-byte aligned (host-checked)
    const long ntile = (ngrid + FK_ROWS - 1) / FK_ROWS;
    const double *c0 = coef, *c1 = coef + (size_t)ngrid, *c2 = coef + 2 * (size_t)ngrid,
                 *c3 = coef + 3 * (size_t)ngrid;

    bool va[2], vb[4];
#pragma unroll
    for (int i = 0; i < 2; ++i) va[i] = wa + 4 * i < NT;
#pragma unroll
    for (int j = 0; j < 4; ++j) vb[j] = wb + 2 * j < NT;

    d4 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (d4){0.0, 0.0, 0.0, 0.0};

    double p0[2 * C::JN], p1[2 * C::JN], p2[2 * C::JN], p3[2 * C::JN];
    double k0 = 0, k1 = 0, k2 = 0, k3 = 0;
    auto fetch = [&](long tt) {
        const long g = tt * FK_ROWS + row;
        const bool ok = tt < ntile && g < ngrid;
        const size_t ro = (size_t)(ok ? g : 0) * nao;
#pragma unroll
        for (int j = 0; j < C::JN; ++j) {
            const int c = 32 * j + 2 * seg;
            load_pair(ao + ro, c, nao, ok, vec, p0[2 * j], p0[2 * j + 1]);
            if (GRAD) {
                load_pair(gx + ro, c, nao, ok, vec, p1[2 * j], p1[2 * j + 1]);
                load_pair(gy + ro, c, nao, ok, vec, p2[2 * j], p2[2 * j + 1]);
                load_pair(gz + ro, c, nao, ok, vec, p3[2 * j], p3[2 * j + 1]);
            }
        }
        k0 = ok ? c0[g] : 0.0;
        if (GRAD) {
            k1 = ok ? c1[g] : 0.0;
            k2 = ok ? c2[g] : 0.0;
            k3 = ok ? c3[g] : 0.0;
        }
    };

    long t = blockIdx.x;
    fetch(t);
    int buf = 0;
    for (; t < ntile; t += gridDim.x, buf ^= 1) {
        double *P = Ps + buf * TILE, *Q = Qs + buf * TILE;
#pragma unroll
        for (int j = 0; j < C::JN; ++j) {
            const int c = 32 * j + 2 * seg;
            if (c < C::NCOL) {
                double qa = k0 * p0[2 * j], qb = k0 * p0[2 * j + 1];
                if (GRAD) {
                    qa += k1 * p1[2 * j] + k2 * p2[2 * j] + k3 * p3[2 * j];
                    qb += k1 * p1[2 * j + 1] + k2 * p2[2 * j + 1] + k3 * p3[2 * j + 1];
                }
                *reinterpret_cast<double2 *>(&Q[row * C::LDX + c]) = make_double2(qa, qb);
                *reinterpret_cast<double2 *>(&P[row * C::LDX + c]) = make_double2(p0[2 * j], p0[2 * j + 1]);
            }
        }
        __syncthreads();
        fetch(t + gridDim.x); // next sub-tile's planes fly during the MFMA phase
#pragma unroll
        for (int ks = 0; ks < FK_ROWS / 4; ++ks) {
            const int o = (4 * ks + lk) * C::LDX + li;
            double af[2], bf[4];
#pragma unroll
            for (int i = 0; i < 2; ++i) af[i] = va[i] ? Q[o + 16 * (wa + 4 * i)] : 0.0;
#pragma unroll
            for (int j = 0; j < 4; ++j) bf[j] = vb[j] ? P[o + 16 * (wb + 2 * j)] : 0.0;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (va[i] && vb[j]) acc[i][j] = mfma_f64(af[i], bf[j], acc[i][j]);
        }
        // double-buffered: the other buffer is rewritten only after the next barrier, which
        // every wave reaches after finishing this MFMA phase.
    }

    double *slab = slabs + (size_t)blockIdx.x * nao * nao;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (!(va[i] && vb[j])) continue;
            const int b = 16 * (wb + 2 * j) + li;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int a = 16 * (wa + 4 * i) + lk + 4 * r;
                if (a < nao && b < nao) slab[(size_t)a * nao + b] = acc[i][j][r];
            }
        }
}

// Sum of the per-workgroup slabs, 4 slab groups per element in parallel, fixed order.
template <bool SYM>
__global__ __launch_bounds__(256) void k_reduce_slabs4(int nao, int nslab,
                                                       const double *__restrict__ slabs,
                                                       double *__restrict__ V)
{
    __shared__ double part[4][64];
    const int el = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int e = blockIdx.x * 64 + el;
    const size_t n2 = (size_t)nao * nao;
    double s = 0.0;
    if (e < (int)n2) {
        const int a = e / nao, b = e - a * nao;
        const int et = b * nao + a;
        for (int k = grp; k < nslab; k += 4) {
            double v = slabs[k * n2 + e];
            if (SYM) v += slabs[k * n2 + et]; // (x + y) == (y + x): V comes out bitwise symmetric
            s += v;
        }
    }
    part[grp][el] = s;
    __syncthreads();
    if (grp == 0 && e < (int)n2) V[e] = (part[0][el] + part[1][el]) + (part[2][el] + part[3][el]);
}

} // namespace qcdft
