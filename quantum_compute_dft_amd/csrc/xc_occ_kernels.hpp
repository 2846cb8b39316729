// Density step through the OCCUPIED ORBITALS (extension DFT_ComputeXCOcc): with dm = C C^T, C (nao x nocc),
//     Y    = AO . C                    (ngrid x nocc)
//     rho  = rowsum(Y * Y)
//     X    = Y . C^T                   (ngrid x nao)
//     grad rho = 2 rowsum(X * dAO)
// instead of X = AO . Ds with the full nao x nao matrix (src/dft_solver.cu:294-307, 346-380 contract with the full
// dm; the driver holds C already, dft.py:181-182).  fp64-MFMA work per grid row: 4 nao nocc flops instead of
// 2 nao^2 -- 0.41x at Benzene/def2-SVP (114, 21), 0.19x at Anthracene/def2-TZVP (494, 47), 0.45x at C33.../def2-SVP
// (1150, 250) -- and the AO plane is read once, for Y only (rho needs no second dot with it).
//
// Shape of the kernel: every WAVE owns 16 grid rows and chains the two products in registers.  The first one is
// formed transposed, Y^T = C^T . AO^T, so that its result registers -- lane (l&15 = grid row, l>>4 = q), register
// r = orbital 16t + 4r + q -- ARE the A-operand fragments of the second (A[i = grid row][k = orbital quad index q]):
// Y never touches LDS.  The second product takes its AO columns in the permuted order nu = 32J + 2j + c (two
// tiles c = 0, 1 per 32-column block J), so that lane (j, q) ends up with X for the column PAIR (32J + 2j, +1) of
// rows q + 4r: exactly the 16-byte (row, seg) pattern -- 16 lanes = 256 contiguous bytes of one grid row -- in which
// the gradient planes stream best (6.1-6.4 TB/s, profiles/r02_stream_pattern_probe2.txt); the row dots run in
// registers against those loads and X never touches LDS either.  LDS holds only
//   * C, zero-padded, as [nu][orbital] with an ODD leading dimension: both fragment reads (phase 1: 2 k-rows x 16
//     orbitals per 32 lanes, k-rows 16 apart; phase 2: 16 even rows x 2 orbitals) hit 32 distinct 8-byte banks;
//   * per wave one 16 x 32 chunk of AO (ld 33) that turns the quad-coalesced global loads (4 lanes = 64
//     contiguous bytes of a row) into B-operand fragments; wave-private, so no barrier guards it.
// RESIDENT: all of C stays in LDS (nao <= 128 and the like), the workgroups are persistent and their waves never
// meet at a barrier after the prologue.  Otherwise C streams through a double-buffered 32-row chunk per step
// (phase 1: k-chunk c, phase 2: column block J -- the same rows of C), one barrier per step.
// More than 16*NTO occupied orbitals are taken in `npass` passes over the planes (rho and the row dots are
// linear in the orbital sum).
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>
#include "device_util.hpp"

namespace qcdft {

constexpr int OC_KC = 32;   // AO columns per chunk (k of phase 1, nu of phase 2)
constexpr int OC_LDA = 33;  // wave-private AO chunk [16][33]

template <int NTO> struct OccCfg {
    static constexpr int NOP = 16 * NTO;       // padded orbitals per pass
    static constexpr int LDC = NOP + 1;        // odd
    static constexpr int CHUNK = OC_KC * LDC;  // doubles of one 32-row chunk of C in LDS
};

// bytes of dynamic LDS the kernel needs
inline size_t occ_lds_bytes(int nto, int nw, bool resident, int nch, int npass)
{
    const size_t chunk = (size_t)OC_KC * (16 * nto + 1);
    return sizeof(double) * ((resident ? (size_t)npass * nch : 2) * chunk + (size_t)nw * 16 * OC_LDA);
}

// cp[pass][nu][o] = C[nu][pass*NOP + o], zero outside nao x nocc; nu < 32*nch
__global__ __launch_bounds__(256) void k_pack_cocc_occ(int nao, int nocc, int nop, int nch, int npass,
                                                       const double *__restrict__ c, double *__restrict__ cp)
{
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    const long per_pass = (long)OC_KC * nch * nop;
    if (e >= per_pass * npass) return;
    const int pass = (int)(e / per_pass);
    const long r = e - (long)pass * per_pass;
    const int nu = (int)(r / nop), o = (int)(r % nop), orb = pass * nop + o;
    cp[e] = (nu < nao && orb < nocc) ? c[(size_t)nu * nocc + orb] : 0.0;
}

// dm = C C^T (only when the caller of DFT_ComputeXCOcc passed no dm and the dm kernels are the better path)
__global__ __launch_bounds__(256) void k_dm_from_cocc(int nao, int nocc, const double *__restrict__ c,
                                                      double *__restrict__ dm)
{
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= (long)nao * nao) return;
    const int a = (int)(e / nao), b = (int)(e % nao);
    const double *ca = c + (size_t)a * nocc, *cb = c + (size_t)b * nocc;
    double s = 0.0;
    for (int i = 0; i < nocc; ++i) s += ca[i] * cb[i];
    dm[e] = s;
}

template <int NTO, int NW, bool GRAD, bool VEC, bool RESIDENT, int GSETS>
__global__ __launch_bounds__(64 * NW, 2) void k_rho_occ(long ngrid, int nao, int nch, int npass,
                                                        const double *__restrict__ ao,
                                                        const double *__restrict__ gx,
                                                        const double *__restrict__ gy,
                                                        const double *__restrict__ gz,
                                                        const double *__restrict__ cp,
                                                        double *__restrict__ rho,
                                                        double *__restrict__ grad,
                                                        double *__restrict__ sigma)
{
    using C = OccCfg<NTO>;
    constexpr int T = 64 * NW;
    constexpr int NPAIR = OC_KC * C::NOP / 2;           // double2 elements of one chunk of C
    constexpr int NL = (NPAIR + T - 1) / T;             // per thread
    extern __shared__ double lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, q = lane >> 4;
    double *const Cs = lds;                                                        // C: resident or two chunks
    const int nchunk_all = npass * nch;
    double *const As = lds + (size_t)(RESIDENT ? nchunk_all : 2) * C::CHUNK + wave * (16 * OC_LDA);

    // ---- staging of C chunks (global, L2-resident, packed [chunk][32][NOP]) into LDS [32][LDC]
    double2 cr[NL];
    auto cp_fetch = [&](int chunk) { // chunk index in [0, npass*nch)
        const double2 *src = reinterpret_cast<const double2 *>(cp + (size_t)chunk * (OC_KC * C::NOP));
#pragma unroll
        for (int j = 0; j < NL; ++j) {
            const int e = tid + j * T;
            cr[j] = e < NPAIR ? src[e] : make_double2(0.0, 0.0);
        }
    };
    auto cp_stash = [&](double *dst) {
#pragma unroll
        for (int j = 0; j < NL; ++j) {
            const int e = tid + j * T;
            if (e < NPAIR) {
                const int nu = (2 * e) / C::NOP, o = (2 * e) % C::NOP;
                dst[nu * C::LDC + o] = cr[j].x;
                dst[nu * C::LDC + o + 1] = cr[j].y;
            }
        }
    };
    if (RESIDENT) {
        for (int ch = 0; ch < nchunk_all; ++ch) {
            cp_fetch(ch);
            cp_stash(Cs + (size_t)ch * C::CHUNK);
        }
        __syncthreads();
    }

    const long plane = ngrid * (long)nao;
    const long nrb = (ngrid + 16 * NW - 1) / (16 * NW);
    // AO chunk loads: lane (row = lane>>2, p = lane&3), four loads m: columns 32c + 8m + 2p, +1
    const int a_row = lane >> 2, a_p = lane & 3;
    const unsigned a_voff = (unsigned)(a_row * nao + 2 * a_p) * 8u;
    // gradient block loads: lane (q, seg = li), rows 4r + q, columns 32J + 2 li, +1
    unsigned g_voff[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) g_voff[r] = (unsigned)((4 * r + q) * nao + 2 * li) * 8u;
    // phase-1 k mapping inside a chunk: k(q, s) = 16 (q&1) + 8 (q>>1) + s, s < 8
    const int kq = 16 * (q & 1) + 8 * (q >> 1);

    for (long rb = blockIdx.x; rb < nrb; rb += gridDim.x) {
        const long row0 = (rb * NW + wave) * 16;          // wave-uniform
        const bool live = row0 < ngrid;
        const long e0 = (live ? row0 : 0) * (long)nao;
        const __amdgpu_buffer_rsrc_t r0 = plane_tile_rsrc(ao, plane, e0, live);
        const __amdgpu_buffer_rsrc_t r0dead = plane_tile_rsrc(ao, plane, e0, false);
        double rho_acc = 0.0;
        double s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0}, s3[4] = {0, 0, 0, 0};

        int step = 0; // streamed mode: chunk steps since the start of this row block (buffer parity)
        if (!RESIDENT) { // first chunk of the block
            cp_fetch(0);
            cp_stash(Cs);
            __syncthreads();
        }
        for (int pass = 0; pass < npass; ++pass) {
            d4 y[NTO];
#pragma unroll
            for (int t = 0; t < NTO; ++t) y[t] = (d4){0.0, 0.0, 0.0, 0.0};

            // ------------------------------------------------ phase 1: Y^T = C^T . AO^T
            double2 av[4];
            auto issue_ao = [&](int c) {
                const __amdgpu_buffer_rsrc_t rr = c < nch ? r0 : r0dead; // past the last chunk: no traffic, same count
                const unsigned soff = (unsigned)(c * OC_KC) * 8u;
#pragma unroll
                for (int m = 0; m < 4; ++m) av[m] = buf_load_pair2<VEC>(rr, a_voff + 64 * m, soff);
            };
            issue_ao(0);
            for (int c = 0; c < nch; ++c) {
                const int chunk = pass * nch + c;
                const double *Cc = RESIDENT ? Cs + (size_t)chunk * C::CHUNK : Cs + (step & 1) * C::CHUNK;
                if (!RESIDENT) {
                    // next step's chunk: the next k-chunk; behind the last one phase 2 restarts at the pass's first chunk
                    // (LDA: the next pass's first); behind the very last step nothing is needed (a repeat, never read)
                    const int nx = c + 1 < nch ? chunk + 1 : GRAD ? pass * nch : (pass + 1 < npass ? chunk + 1 : chunk);
                    cp_fetch(nx);
                }
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    As[a_row * OC_LDA + 8 * m + 2 * a_p] = av[m].x;
                    As[a_row * OC_LDA + 8 * m + 2 * a_p + 1] = av[m].y;
                }
                __builtin_amdgcn_wave_barrier();
                issue_ao(c + 1);
#pragma unroll
                for (int s = 0; s < 8; ++s) {
                    const double b = As[li * OC_LDA + kq + s];
#pragma unroll
                    for (int t = 0; t < NTO; ++t) y[t] = mfma_f64(Cc[(kq + s) * C::LDC + 16 * t + li], b, y[t]);
                }
                __builtin_amdgcn_wave_barrier();
                if (!RESIDENT) {
                    cp_stash(Cs + ((step + 1) & 1) * C::CHUNK);
                    __syncthreads();
                    ++step;
                }
            }
            {
                double loc = 0.0;
#pragma unroll
                for (int t = 0; t < NTO; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) loc += y[t][r] * y[t][r];
                rho_acc += loc;
            }

            // ------------------------------------------------ phase 2: X = Y . C^T by 32-column blocks, row dots
            if (GRAD) {
                const __amdgpu_buffer_rsrc_t r1 = plane_tile_rsrc(gx, plane, e0, live);
                const __amdgpu_buffer_rsrc_t r2 = plane_tile_rsrc(gy, plane, e0, live);
                const __amdgpu_buffer_rsrc_t r3 = plane_tile_rsrc(gz, plane, e0, live);
                const __amdgpu_buffer_rsrc_t r1d = plane_tile_rsrc(gx, plane, e0, false);
                const __amdgpu_buffer_rsrc_t r2d = plane_tile_rsrc(gy, plane, e0, false);
                const __amdgpu_buffer_rsrc_t r3d = plane_tile_rsrc(gz, plane, e0, false);
                double2 gv[GSETS][3][4];
                auto issue_g = [&](auto S, int J) {
                    constexpr int st = decltype(S)::value;
                    const bool in = J < nch;
                    const __amdgpu_buffer_rsrc_t a = in ? r1 : r1d, b = in ? r2 : r2d, c = in ? r3 : r3d;
                    const unsigned soff = (unsigned)(J * OC_KC) * 8u;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        gv[st][0][r] = buf_load_pair2<VEC>(a, g_voff[r], soff);
                        gv[st][1][r] = buf_load_pair2<VEC>(b, g_voff[r], soff);
                        gv[st][2][r] = buf_load_pair2<VEC>(c, g_voff[r], soff);
                    }
                };
                auto block = [&](auto S, int J, int Jnext) { // MFMAs of block J, then its row dots against set S
                    constexpr int st = decltype(S)::value;
                    const int chunk = pass * nch + J;
                    const double *Cc = RESIDENT ? Cs + (size_t)chunk * C::CHUNK : Cs + (step & 1) * C::CHUNK;
                    if (!RESIDENT) {
                        // next step: block J+1 of this pass, or the first k-chunk of the next pass
                        const int nx = J + 1 < nch ? chunk + 1 : (pass + 1 < npass ? (pass + 1) * nch : chunk);
                        cp_fetch(nx);
                    }
                    if (GSETS == 1) issue_g(S, J); // one register set: the loads fly under this block's own MFMAs
                    else if (Jnext >= 0) issue_g(std::integral_constant<int, (st + 1) % GSETS>{}, Jnext);
                    d4 xe = (d4){0.0, 0.0, 0.0, 0.0}, xo = (d4){0.0, 0.0, 0.0, 0.0};
                    const double *Ce = Cc + (2 * li) * C::LDC + q, *Co = Ce + C::LDC;
#pragma unroll
                    for (int t = 0; t < NTO; ++t)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            xe = mfma_f64(y[t][r], Ce[16 * t + 4 * r], xe);
                            xo = mfma_f64(y[t][r], Co[16 * t + 4 * r], xo);
                        }
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        s1[r] += xe[r] * gv[st][0][r].x + xo[r] * gv[st][0][r].y;
                        s2[r] += xe[r] * gv[st][1][r].x + xo[r] * gv[st][1][r].y;
                        s3[r] += xe[r] * gv[st][2][r].x + xo[r] * gv[st][2][r].y;
                    }
                    if (!RESIDENT) {
                        cp_stash(Cs + ((step + 1) & 1) * C::CHUNK);
                        __syncthreads();
                        ++step;
                    }
                };
                using S0 = std::integral_constant<int, 0>;
                using S1 = std::integral_constant<int, GSETS - 1>;
                if (GSETS == 1) {
                    for (int J = 0; J < nch; ++J) block(S0{}, J, -1);
                } else {
                    issue_g(S0{}, 0);
                    int J = 0;
                    for (; J + 1 < nch; J += 2) {
                        block(S0{}, J, J + 1);
                        block(S1{}, J + 1, J + 2); // J + 2 == nch: a dead issue keeps the load count static
                    }
                    if (J < nch) block(S0{}, J, -1);
                }
            }
        }

        // ---- rows of this wave: rho from lanes (li = row), gradient sums from lanes (q, r): rows q + 4r
        double rt = rho_acc;
        rt += __shfl_xor(rt, 16, 64);
        rt += __shfl_xor(rt, 32, 64);
        if (live) {
            if (lane < 16 && row0 + lane < ngrid) rho[row0 + lane] = rt;
            if (GRAD) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const double ax = 2.0 * row16_sum(s1[r]), ay = 2.0 * row16_sum(s2[r]), az = 2.0 * row16_sum(s3[r]);
                    const long g = row0 + q + 4 * r;
                    if (li == 0 && g < ngrid) {
                        grad[3 * g + 0] = ax;
                        grad[3 * g + 1] = ay;
                        grad[3 * g + 2] = az;
                        sigma[g] = ax * ax + ay * ay + az * az;
                    }
                }
            }
        }
    }
}

} // namespace qcdft
