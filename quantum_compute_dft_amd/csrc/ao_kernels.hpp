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
// Mapping: workgroup = 64 grid points (lane = point), the four waves split the
// shells of one <=32-column chunk; results go to an LDS tile [point][col]
// (ld 33: conflict-free both for lane=point writes and lane=column reads) and
// leave as contiguous row segments.  Shell parameters are wave-uniform
// (scalar loads).  HBM-write bound: 8*nao*(1 or 4) bytes per grid point.
#pragma once
#include <hip/hip_runtime.h>

namespace qcdft {

constexpr int AO_MAX_L = 3;
constexpr int AO_CT = 32;  // columns per chunk
constexpr int AO_LD = 33;  // LDS leading dimension
constexpr int AO_G = 64;   // grid points per workgroup

struct AoShell {
    double x, y, z;
    int l, nprim, off, ao;
};
struct AoChunk {
    int shell_lo, shell_hi, col_lo, ncol;
};

template <bool GRAD>
__device__ __forceinline__ void ao_put(double *tile, int idx, double S, double Sx, double Sy,
                                       double Sz, double R0, double R1, double dx, double dy,
                                       double dz)
{
    tile[idx] = R0 * S;
    if (GRAD) {
        const double t = R1 * S;
        tile[AO_G * AO_LD + idx] = R0 * Sx + t * dx;
        tile[2 * AO_G * AO_LD + idx] = R0 * Sy + t * dy;
        tile[3 * AO_G * AO_LD + idx] = R0 * Sz + t * dz;
    }
}

template <bool GRAD>
__global__ __launch_bounds__(256) void k_eval_ao(long ngrid, int nao, int nchunk,
                                                 const AoShell *__restrict__ sh,
                                                 const double *__restrict__ pexp,
                                                 const double *__restrict__ pcoef,
                                                 const AoChunk *__restrict__ chunks,
                                                 const double *__restrict__ coords,
                                                 double *__restrict__ ao,
                                                 double *__restrict__ grad)
{
    __shared__ double tile[(GRAD ? 4 : 1) * AO_G * AO_LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long g0 = (long)blockIdx.x * AO_G;
    const long g = g0 + lane;
    double px = 0, py = 0, pz = 0;
    if (g < ngrid) {
        px = coords[3 * g];
        py = coords[3 * g + 1];
        pz = coords[3 * g + 2];
    }
    for (int ci = 0; ci < nchunk; ++ci) {
        const AoChunk ch = chunks[ci];
        for (int s = ch.shell_lo + wave; s < ch.shell_hi; s += 4) {
            const AoShell q = sh[s];
            const double x = px - q.x, y = py - q.y, z = pz - q.z;
            const double r2 = x * x + y * y + z * z;
            double R0 = 0.0, R1 = 0.0;
            for (int p = 0; p < q.nprim; ++p) {
                const double a = pexp[q.off + p];
                const double e = pcoef[q.off + p] * exp(-a * r2);
                R0 += e;
                R1 -= 2.0 * a * e;
            }
            const int i0 = lane * AO_LD + (q.ao - ch.col_lo);
            if (q.l == 0) {
                constexpr double c = 0.282094791773878143;
                ao_put<GRAD>(tile, i0, c, 0, 0, 0, R0, R1, x, y, z);
            } else if (q.l == 1) {
                constexpr double c = 0.488602511902919921;
                ao_put<GRAD>(tile, i0 + 0, c * x, c, 0, 0, R0, R1, x, y, z);
                ao_put<GRAD>(tile, i0 + 1, c * y, 0, c, 0, R0, R1, x, y, z);
                ao_put<GRAD>(tile, i0 + 2, c * z, 0, 0, c, R0, R1, x, y, z);
            } else if (q.l == 2) {
                constexpr double c = 1.092548430592079070, d = 0.315391565252520002,
                                 e = 0.546274215296039535;
                ao_put<GRAD>(tile, i0 + 0, c * x * y, c * y, c * x, 0, R0, R1, x, y, z);
                ao_put<GRAD>(tile, i0 + 1, c * y * z, 0, c * z, c * y, R0, R1, x, y, z);
                ao_put<GRAD>(tile, i0 + 2, d * (2 * z * z - x * x - y * y), -2 * d * x, -2 * d * y,
                             4 * d * z, R0, R1, x, y, z);
                ao_put<GRAD>(tile, i0 + 3, c * x * z, c * z, 0, c * x, R0, R1, x, y, z);
                ao_put<GRAD>(tile, i0 + 4, e * (x * x - y * y), 2 * e * x, -2 * e * y, 0, R0, R1, x,
                             y, z);
            } else {
                constexpr double f3 = 0.590043589926643510, f2 = 2.890611442640554055,
                                 f1 = 0.457045799464465739, f0 = 0.373176332590115391,
                                 f2b = 1.445305721320277020;
                const double xx = x * x, yy = y * y, zz = z * z;
                ao_put<GRAD>(tile, i0 + 0, f3 * y * (3 * xx - yy), f3 * 6 * x * y,
                             f3 * (3 * xx - 3 * yy), 0, R0, R1, x, y, z);
                ao_put<GRAD>(tile, i0 + 1, f2 * x * y * z, f2 * y * z, f2 * x * z, f2 * x * y, R0,
                             R1, x, y, z);
                ao_put<GRAD>(tile, i0 + 2, f1 * y * (4 * zz - xx - yy), -2 * f1 * x * y,
                             f1 * (4 * zz - xx - 3 * yy), 8 * f1 * y * z, R0, R1, x, y, z);
                ao_put<GRAD>(tile, i0 + 3, f0 * z * (2 * zz - 3 * xx - 3 * yy), -6 * f0 * x * z,
                             -6 * f0 * y * z, f0 * (6 * zz - 3 * xx - 3 * yy), R0, R1, x, y, z);
                ao_put<GRAD>(tile, i0 + 4, f1 * x * (4 * zz - xx - yy), f1 * (4 * zz - 3 * xx - yy),
                             -2 * f1 * x * y, 8 * f1 * x * z, R0, R1, x, y, z);
                ao_put<GRAD>(tile, i0 + 5, f2b * z * (xx - yy), 2 * f2b * x * z, -2 * f2b * y * z,
                             f2b * (xx - yy), R0, R1, x, y, z);
                ao_put<GRAD>(tile, i0 + 6, f3 * x * (xx - 3 * yy), f3 * (3 * xx - 3 * yy),
                             -6 * f3 * x * y, 0, R0, R1, x, y, z);
            }
        }
        __syncthreads();
        {
            const int c = tid & 31, r0 = tid >> 5;
            if (c < ch.ncol) {
#pragma unroll
                for (int p = 0; p < 8; ++p) {
                    const int r = r0 + 8 * p;
                    const long gr = g0 + r;
                    if (gr < ngrid) {
                        const size_t o = (size_t)gr * nao + ch.col_lo + c;
                        ao[o] = tile[r * AO_LD + c];
                        if (GRAD) {
                            const size_t plane = (size_t)ngrid * nao;
                            grad[o] = tile[AO_G * AO_LD + r * AO_LD + c];
                            grad[plane + o] = tile[2 * AO_G * AO_LD + r * AO_LD + c];
                            grad[2 * plane + o] = tile[3 * AO_G * AO_LD + r * AO_LD + c];
                        }
                    }
                }
            }
        }
        __syncthreads();
    }
}

inline void launch_eval_ao(hipStream_t st, long ngrid, int nao, int nchunk, const AoShell *sh,
                           const double *pexp, const double *pcoef, const AoChunk *chunks,
                           const double *coords, double *ao, double *grad)
{
    dim3 g((unsigned)((ngrid + AO_G - 1) / AO_G));
    if (grad)
        hipLaunchKernelGGL(k_eval_ao<true>, g, dim3(256), 0, st, ngrid, nao, nchunk, sh, pexp, pcoef, chunks, coords, ao, grad);
    else
        hipLaunchKernelGGL(k_eval_ao<false>, g, dim3(256), 0, st, ngrid, nao, nchunk, sh, pexp, pcoef, chunks, coords, ao, grad);
}

} // namespace qcdft
