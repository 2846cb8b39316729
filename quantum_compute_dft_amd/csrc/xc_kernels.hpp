// HIP kernels of the XC sweep for gfx950 (MI355X): density/gradient contraction,
// pointwise XC, Vxc accumulation.  fp64 throughout; contractions run on the
// fp64 matrix cores (v_mfma_f64_16x16x4_f64).
//
// Replaces the reference's thread-per-grid-point kernels (src/dft_solver.cu):
//   get_rho_kernel :294-307, get_rho_sigma_kernel_planar :346-380   -> k_rho_*
//   lda/gga/b3lyp_fused_kernel :309-344,:382-432,:434-513 (both passes)
//   + reduce_sum_kernel :285-292                                    -> k_xc_points, k_reduce_slabs8 (last block)
//   B matrix + cublasDgemm :541-548,:580,:616,:663                  -> k_vxc_* (B never materialised)
//   symmetrize_matrix_kernel :515-527                               -> k_reduce_slabs8<true> (xc_ws_kernels.hpp)
//
// Formulation (identical result, different arithmetic order):
//   Ds  = (D + D^T)/2                       (exact for any D: rho and grad rho only see the symmetric part)
//   X   = AO . Ds                           (ngrid x nao, fp64 MFMA, never stored)
//   rho = rowsum(X * AO), grad rho = 2 rowsum(X * dAO)
//   c0..c3 per grid point from the functional (xc_functionals.hpp)
//   V[a][b] = sum_g (c0 AO + c1 dxAO + c2 dyAO + c3 dzAO)[g][a] * AO[g][b]   (fp64 MFMA, split over grid chunks)
//
// fp64 MFMA operand maps (cdna_hip_programming.md section 3): lane l holds
// A[i = l&15][k = l>>4], B[k = l>>4][j = l&15]; result register r holds
// D[row = (l>>4) + 4r][col = l&15].
#pragma once
#include <hip/hip_runtime.h>
#include "device_util.hpp"
#include "xc_functionals.hpp"

namespace qcdft {


// Ds = (D + D^T)/2, zero-padded to NP x NP (NP = 16*ceil(nao/16)).
__global__ void k_sym_dm(int nao, int NP, const double *__restrict__ D, double *__restrict__ Dp)
{
    const int i = blockIdx.y * 16 + threadIdx.y, j = blockIdx.x * 16 + threadIdx.x;
    if (i >= NP || j >= NP) return;
    double v = 0.0;
    if (i < nao && j < nao) v = 0.5 * (D[(size_t)i * nao + j] + D[(size_t)j * nao + i]);
    Dp[(size_t)i * NP + j] = v;
}

// ---------------------------------------------------------------- rho ------
// One workgroup = 64 grid points (one 16-row MFMA block per wave), all AO
// columns in chunks of 128.  AO k-slab and Ds tile staged in LDS.
//   As[m][k] ld 34  : A-operand reads (16 rows x 2 k per 32 lanes) hit 32 distinct bank pairs
//   Bs[k][n] ld 144 : B-operand reads (2 k-rows x 16 cols per 32 lanes) likewise
template <bool GRAD>
__global__ __launch_bounds__(256) void k_rho_mfma(long ngrid, int nao, int NP,
                                                  const double *__restrict__ ao,
                                                  const double *__restrict__ gx,
                                                  const double *__restrict__ gy,
                                                  const double *__restrict__ gz,
                                                  const double *__restrict__ Dp,
                                                  double *__restrict__ rho,
                                                  double *__restrict__ grad,
                                                  double *__restrict__ sigma)
{
    constexpr int BM = 64, BK = 32, BN = 128, LDA = BK + 2, LDB = BN + 16;
    __shared__ double As[BM * LDA];
    __shared__ double Bs[BK * LDB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lk = lane >> 4;
    const long g0 = (long)blockIdx.x * BM;

    double s0[4] = {0, 0, 0, 0}, s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0}, s3[4] = {0, 0, 0, 0};

    for (int n0 = 0; n0 < NP; n0 += BN) {
        const int nt = min(8, (NP - n0) >> 4);
        d4 acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = (d4){0.0, 0.0, 0.0, 0.0};

        for (int k0 = 0; k0 < NP; k0 += BK) {
            {
                const int c = tid & 31, r0 = tid >> 5, k = k0 + c;
#pragma unroll
                for (int p = 0; p < 8; ++p) {
                    const int r = r0 + 8 * p;
                    const long g = g0 + r;
                    As[r * LDA + c] = (g < ngrid && k < nao) ? ao[(size_t)g * nao + k] : 0.0;
                }
            }
            {
                const int c = tid & 127, r0 = tid >> 7, n = n0 + c;
#pragma unroll
                for (int p = 0; p < 16; ++p) {
                    const int r = r0 + 2 * p, k = k0 + r;
                    Bs[r * LDB + c] = (k < NP && n < NP) ? Dp[(size_t)k * NP + n] : 0.0;
                }
            }
            __syncthreads();
            const int nks = min(BK, NP - k0) >> 2;
            for (int ks = 0; ks < nks; ++ks) {
                const double a = As[(wave * 16 + li) * LDA + ks * 4 + lk];
                const double *brow = &Bs[(ks * 4 + lk) * LDB + li];
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    if (j < nt) acc[j] = mfma_f64(a, brow[j * 16], acc[j]);
            }
            __syncthreads();
        }

        // X tile (16 rows x 128 cols per wave) times the AO planes, row-wise
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int col = n0 + 16 * j + li;
            if (j < nt && col < nao) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const long g = g0 + wave * 16 + lk + 4 * r;
                    if (g < ngrid) {
                        const size_t idx = (size_t)g * nao + col;
                        const double x = acc[j][r];
                        s0[r] += x * ao[idx];
                        if (GRAD) {
                            s1[r] += x * gx[idx];
                            s2[r] += x * gy[idx];
                            s3[r] += x * gz[idx];
                        }
                    }
                }
            }
        }
    }

    // sum over the 16 lanes that share a row (xor butterfly: every lane gets the total)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
        for (int m = 1; m < 16; m <<= 1) {
            s0[r] += __shfl_xor(s0[r], m, 64);
            if (GRAD) {
                s1[r] += __shfl_xor(s1[r], m, 64);
                s2[r] += __shfl_xor(s2[r], m, 64);
                s3[r] += __shfl_xor(s3[r], m, 64);
            }
        }
    }
    if (li < 4) { // lane li of each 16-lane group writes row register li
        const int r = li;
        const long g = g0 + wave * 16 + lk + 4 * r;
        if (g < ngrid) {
            double v0 = s0[0], v1 = s1[0], v2 = s2[0], v3 = s3[0];
            if (r == 1) { v0 = s0[1]; v1 = s1[1]; v2 = s2[1]; v3 = s3[1]; }
            if (r == 2) { v0 = s0[2]; v1 = s1[2]; v2 = s2[2]; v3 = s3[2]; }
            if (r == 3) { v0 = s0[3]; v1 = s1[3]; v2 = s2[3]; v3 = s3[3]; }
            rho[g] = v0;
            if (GRAD) {
                const double ax = 2.0 * v1, ay = 2.0 * v2, az = 2.0 * v3;
                grad[3 * g + 0] = ax;
                grad[3 * g + 1] = ay;
                grad[3 * g + 2] = az;
                sigma[g] = ax * ax + ay * ay + az * az;
            }
        }
    }
}

// Plain-VALU validation kernel of the same contraction: one wave per grid point.
template <bool GRAD>
__global__ __launch_bounds__(256) void k_rho_valu(long ngrid, int nao, int NP,
                                                  const double *__restrict__ ao,
                                                  const double *__restrict__ gx,
                                                  const double *__restrict__ gy,
                                                  const double *__restrict__ gz,
                                                  const double *__restrict__ Dp,
                                                  double *__restrict__ rho,
                                                  double *__restrict__ grad,
                                                  double *__restrict__ sigma)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long g = (long)blockIdx.x * 4 + wave;
    if (g >= ngrid) return;
    const double *phi = ao + (size_t)g * nao;
    double r = 0, sx = 0, sy = 0, sz = 0;
    for (int v = lane; v < nao; v += 64) {
        double x = 0.0;
        for (int u = 0; u < nao; ++u) x += Dp[(size_t)u * NP + v] * phi[u];
        r += x * phi[v];
        if (GRAD) {
            sx += x * gx[(size_t)g * nao + v];
            sy += x * gy[(size_t)g * nao + v];
            sz += x * gz[(size_t)g * nao + v];
        }
    }
    for (int m = 32; m >= 1; m >>= 1) {
        r += __shfl_xor(r, m, 64);
        if (GRAD) {
            sx += __shfl_xor(sx, m, 64);
            sy += __shfl_xor(sy, m, 64);
            sz += __shfl_xor(sz, m, 64);
        }
    }
    if (lane == 0) {
        rho[g] = r;
        if (GRAD) {
            grad[3 * g] = 2 * sx; grad[3 * g + 1] = 2 * sy; grad[3 * g + 2] = 2 * sz;
            sigma[g] = 4 * (sx * sx + sy * sy + sz * sz);
        }
    }
}

// ------------------------------------------------------- pointwise XC ------
// TYPE 0 LDA, 1 GGA(PBE), 2 B3LYP.  coef is SoA: c0[ngrid], c1[ngrid], ...
// Each block leaves one deterministic partial of sum_g w_g exc_g.
template <int TYPE>
__global__ __launch_bounds__(256) void k_xc_points(long ngrid, const double *__restrict__ rho,
                                                   const double *__restrict__ sigma,
                                                   const double *__restrict__ grad,
                                                   const double *__restrict__ w,
                                                   double *__restrict__ coef,
                                                   double *__restrict__ partial, int quirks)
{
    __shared__ double red[4];
    const long g = (long)blockIdx.x * 256 + threadIdx.x;
    double e = 0.0;
    if (g < ngrid) {
        const double wt = w[g], r = rho[g];
        xc::PointXC p;
        if (TYPE == 0) {
            p = xc::lda_point(r, wt, quirks != 0);
        } else {
            const double ax = grad[3 * g], ay = grad[3 * g + 1], az = grad[3 * g + 2];
            if (TYPE == 1) p = xc::gga_point(r, sigma[g], ax, ay, az, wt, quirks != 0);
            else           p = xc::b3lyp_point(r, sigma[g], ax, ay, az, wt);
        }
        coef[g] = p.c0;
        if (TYPE != 0) {
            coef[(size_t)ngrid + g] = p.c1;
            coef[2 * (size_t)ngrid + g] = p.c2;
            coef[3 * (size_t)ngrid + g] = p.c3;
        }
        e = wt * p.exc;
    }
    for (int m = 32; m >= 1; m >>= 1) e += __shfl_down(e, m, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = e;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// ---------------------------------------------------------------- Vxc ------
// blockIdx.x = grid chunk, .y/.z = 128-wide blocks of the a / b index.
// Per 32 grid points: Q[g][a] = sum_c coef_c[g] * plane_c[g][a] and P[g][b] = AO[g][b]
// are staged in LDS (ld 144, conflict-free for both operand reads); the four
// waves own 64x64 quadrants of the 128x128 output block (4x4 MFMA tiles each).
template <bool GRAD>
__global__ __launch_bounds__(256) void k_vxc_mfma(long ngrid, int nao, long chunk,
                                                  const double *__restrict__ ao,
                                                  const double *__restrict__ gx,
                                                  const double *__restrict__ gy,
                                                  const double *__restrict__ gz,
                                                  const double *__restrict__ coef,
                                                  double *__restrict__ slabs)
{
    constexpr int G = 32, LD = 144;
    __shared__ double Ps[G * LD];
    __shared__ double Qs[G * LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lk = lane >> 4;
    const int wa = wave >> 1, wb = wave & 1;
    const int a0 = blockIdx.y * 128, b0 = blockIdx.z * 128;
    const long glo = (long)blockIdx.x * chunk;
    const long ghi = min(ngrid, glo + chunk);

    bool act_a[4], act_b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        act_a[i] = a0 + (4 * wa + i) * 16 < nao;
        act_b[i] = b0 + (4 * wb + i) * 16 < nao;
    }
    d4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (d4){0.0, 0.0, 0.0, 0.0};

    const double *c0 = coef, *c1 = coef + (size_t)ngrid, *c2 = coef + 2 * (size_t)ngrid,
                 *c3 = coef + 3 * (size_t)ngrid;

    for (long gt = glo; gt < ghi; gt += G) {
        {
            const int c = tid & 127, r0 = tid >> 7;
            const int ca = a0 + c, cb = b0 + c;
#pragma unroll 4
            for (int p = 0; p < 16; ++p) {
                const int r = r0 + 2 * p;
                const long g = gt + r;
                double q = 0.0, pv = 0.0;
                if (g < ghi) {
                    if (ca < nao) {
                        const size_t idx = (size_t)g * nao + ca;
                        q = c0[g] * ao[idx];
                        if (GRAD) q += c1[g] * gx[idx] + c2[g] * gy[idx] + c3[g] * gz[idx];
                    }
                    if (cb < nao) pv = ao[(size_t)g * nao + cb];
                }
                Qs[r * LD + c] = q;
                Ps[r * LD + c] = pv;
            }
        }
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < G / 4; ++ks) {
            const int row = (ks * 4 + lk) * LD + li;
            double af[4], bf[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                af[i] = Qs[row + (4 * wa + i) * 16];
                bf[i] = Ps[row + (4 * wb + i) * 16];
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (act_a[i] && act_b[j]) acc[i][j] = mfma_f64(af[i], bf[j], acc[i][j]);
        }
        __syncthreads();
    }

    double *slab = slabs + (size_t)blockIdx.x * nao * nao;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int b = b0 + (4 * wb + j) * 16 + li;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int a = a0 + (4 * wa + i) * 16 + lk + 4 * r;
                if (a < nao && b < nao) slab[(size_t)a * nao + b] = acc[i][j][r];
            }
        }
}

// Plain-VALU validation kernel of the same contraction (one block per grid chunk).
template <bool GRAD>
__global__ __launch_bounds__(256) void k_vxc_valu(long ngrid, int nao, long chunk,
                                                  const double *__restrict__ ao,
                                                  const double *__restrict__ gx,
                                                  const double *__restrict__ gy,
                                                  const double *__restrict__ gz,
                                                  const double *__restrict__ coef,
                                                  double *__restrict__ slabs)
{
    const long glo = (long)blockIdx.x * chunk, ghi = min(ngrid, glo + chunk);
    const double *c0 = coef, *c1 = coef + (size_t)ngrid, *c2 = coef + 2 * (size_t)ngrid,
                 *c3 = coef + 3 * (size_t)ngrid;
    double *slab = slabs + (size_t)blockIdx.x * nao * nao;
    for (int e = threadIdx.x; e < nao * nao; e += 256) {
        const int a = e / nao, b = e - a * nao;
        double s = 0.0;
        for (long g = glo; g < ghi; ++g) {
            const size_t ia = (size_t)g * nao + a;
            double q = c0[g] * ao[ia];
            if (GRAD) q += c1[g] * gx[ia] + c2[g] * gy[ia] + c3[g] * gz[ia];
            s += q * ao[(size_t)g * nao + b];
        }
        slab[e] = s;
    }
}

} // namespace qcdft
