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
//   * per wave one 16 x 32 chunk of AO (ld 33) that turns the coalesced global loads (the same (row, seg)
//     pattern) into B-operand fragments; wave-private, so no barrier guards it.
// RESIDENT: all of C stays in LDS (nao <= 128 and the like), the workgroups are persistent and their waves never
// meet at a barrier after the prologue.  Otherwise C streams through a double-buffered 32-row chunk per step
// (phase 1: k-chunk c, phase 2: column block J -- the same rows of C), one barrier per step.
// More than 16*NTO occupied orbitals are taken in `npass` passes over the planes (rho and the row dots are
// linear in the orbital sum).
#pragma once
#include <hip/hip_runtime.h>
#include <algorithm>
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
    // plane loads, AO chunks and gradient blocks alike: lane (q, seg = li), rows 4r + q, columns 32c + 2 li, +1 --
    // 16 lanes = 256 contiguous bytes of one grid row, four rows per instruction
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
                for (int r = 0; r < 4; ++r) av[r] = buf_load_pair2<VEC>(rr, g_voff[r], soff);
                __builtin_amdgcn_sched_barrier(0); // all loads of a group leave together (see issue_g)
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
                for (int r = 0; r < 4; ++r) {
                    As[(4 * r + q) * OC_LDA + 2 * li] = av[r].x;
                    As[(4 * r + q) * OC_LDA + 2 * li + 1] = av[r].y;
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
                    __builtin_amdgcn_sched_barrier(0);
                    const bool in = J < nch;
                    const __amdgpu_buffer_rsrc_t a = in ? r1 : r1d, b = in ? r2 : r2d, c = in ? r3 : r3d;
                    const unsigned soff = (unsigned)(J * OC_KC) * 8u;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        gv[st][0][r] = buf_load_pair2<VEC>(a, g_voff[r], soff);
                        gv[st][1][r] = buf_load_pair2<VEC>(b, g_voff[r], soff);
                        gv[st][2][r] = buf_load_pair2<VEC>(c, g_voff[r], soff);
                    }
                    // Pinned: left to itself the scheduler trades these twelve loads in flight for registers (it
                    // interleaved them one by one with their waits to reach four waves per SIMD: 106 -> 126 us)
                    __builtin_amdgcn_sched_barrier(0);
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
                    __builtin_amdgcn_sched_barrier(0); // both tiles' MFMAs run under the loads' latency, the waits come after
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

// ------------------------------------------------------------------------------------------------------------------
// Row-shared form for a resident C (nao <= ~250 with nocc <= 64): the FOUR WAVES OF A WORKGROUP SHARE 16 GRID ROWS and
// split their columns -- wave w takes k-chunk w of phase 1 and column block w of phase 2 (the same 32 rows of C) --
// so that at any moment a workgroup reads whole, consecutive grid rows.  That is what HBM rewards
// (tools/stream_pattern_probe3.hip, profiles/r03_stream_pattern_probe3.txt: the four planes of Benzene/def2-SVP in
// 85 us = 6.2 TB/s this way against 96-107 us when every wave walks the 32-column blocks of its own 16 rows, and the
// more workgroups per CU the slower the latter).  Price: the partial Y^T of the four k-chunks meet in LDS (each wave
// stores its result registers as they are, every wave reads the four copies of its own lane's slots back and adds:
// the lane <-> element map of the MFMA result IS the A-operand map of phase 2), two barriers per 16 rows, and the
// per-row gradient sums of the four column blocks are added by wave 0 one tile later.
// Results leave in BURSTS: the per-row outputs (rho, grad rho, sigma: 40 bytes per grid row, 1 % of the traffic) are
// collected in LDS and written every OC_OT tiles by the whole workgroup.  Stored tile by tile they cost 10 us of 85
// (tools/stream_pattern_probe4.hip, profiles/r03_stream_pattern_probe4.txt: "D + stores") -- a trickle of writes keeps
// turning the HBM channels around under the read stream.
constexpr int OC_OT = 16, OC_OR = OC_OT + 1; // tiles per burst, ring slots

template <int NTO, bool GRAD, bool VEC, bool MULTI>
__global__ __launch_bounds__(256, 2) void k_rho_occ_rs(long ngrid, int nao, int nch,
                                                       const double *__restrict__ ao,
                                                       const double *__restrict__ gx,
                                                       const double *__restrict__ gy,
                                                       const double *__restrict__ gz,
                                                       const double *__restrict__ cocc, int nocc,
                                                       double *__restrict__ rho,
                                                       double *__restrict__ grad,
                                                       double *__restrict__ sigma)
{
    using C = OccCfg<NTO>;
    constexpr int YW = NTO * 256 > 16 * OC_LDA ? NTO * 256 : 16 * OC_LDA; // doubles per wave: AO chunk, then Y partial
    extern __shared__ __attribute__((aligned(32))) double lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, q = lane >> 4;
    double *const Cs = lds;
    double *const region = lds + (size_t)nch * C::CHUNK;
    double *const As = region + wave * YW;
    double *const Gs = region + 4 * YW; // [2][4 waves][16 rows][3]
    double *const Ob = Gs + 2 * 4 * 16 * 3; // [OC_OR slots][16 rows][4]: rho, 2 sum_x, 2 sum_y, 2 sum_z

    const long plane = ngrid * (long)nao;
    const long ntile = (ngrid + 15) / 16;
    unsigned g_voff[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) g_voff[r] = (unsigned)((4 * r + q) * nao + 2 * li) * 8u;
    const int kq = 16 * (q & 1) + 8 * (q >> 1);
    const bool mine = wave < nch; // this wave owns chunk / block `wave` (nao <= 96 leaves waves without one)

    // One group of each kind is ALWAYS in flight per wave: the AO chunk of the next tile is issued as soon as this
    // tile's has been staged, the gradient block of the next tile as soon as this tile's has been consumed -- the
    // registers are free at those points, so the prefetch costs none, and every wait is a counted one (AO groups
    // and gradient groups alternate in issue order).  Dead groups (past the grid, or a wave without a block) go
    // through a zero-record descriptor: no traffic, same count.
    double2 av[4], gv[3][4];
    auto issue_ao = [&](double2 (&dst)[4], long tile, int c) {
        const bool in = tile < ntile && c < nch;
        const __amdgpu_buffer_rsrc_t rr = plane_tile_rsrc(ao, plane, (in ? tile : 0) * 16 * (long)nao, in);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int r = 0; r < 4; ++r) dst[r] = buf_load_pair2<VEC>(rr, g_voff[r], (unsigned)(c * OC_KC) * 8u);
        __builtin_amdgcn_sched_barrier(0);
    };
    auto issue_g = [&](double2 (&dst)[3][4], long tile, int J) {
        const bool in = tile < ntile && J < nch;
        const long e0 = (in ? tile : 0) * 16 * (long)nao;
        const __amdgpu_buffer_rsrc_t a = plane_tile_rsrc(gx, plane, e0, in), b = plane_tile_rsrc(gy, plane, e0, in),
                                     c = plane_tile_rsrc(gz, plane, e0, in);
        const unsigned soff = (unsigned)(J * OC_KC) * 8u;
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            dst[0][r] = buf_load_pair2<VEC>(a, g_voff[r], soff);
            dst[1][r] = buf_load_pair2<VEC>(b, g_voff[r], soff);
            dst[2][r] = buf_load_pair2<VEC>(c, g_voff[r], soff);
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    auto finish_rows = [&](int jt, int pb) { // wave 0, lanes 0..15: gradient of the rows of local tile `jt` from the four partials
        if (lane < 16) {
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const double *p = Gs + ((pb * 4) * 16 + lane) * 3 + k;
                Ob[((jt % OC_OR) * 16 + lane) * 4 + 1 + k] = 2.0 * ((p[0] + p[48]) + (p[96] + p[144]));
            }
        }
    };
    auto flush = [&](int j0, int j1) { // local tiles [j0, j1) (at most OC_OT): thread = (tile, row)
        const int j = j0 + (tid >> 4), row = tid & 15;
        const long g = ((long)blockIdx.x + (long)j * gridDim.x) * 16 + row;
#ifdef QCDFT_OCC_NO_STORE // ablation build only (tools/occ_ablate.hip)
        if (j < j1 && g < 0) {
#else
        if (j < j1 && g < ngrid) {
#endif
            const double *o = Ob + ((j % OC_OR) * 16 + row) * 4;
            rho[g] = o[0];
            if (GRAD) {
                const double ax = o[1], ay = o[2], az = o[3];
                grad[3 * g + 0] = ax;
                grad[3 * g + 1] = ay;
                grad[3 * g + 2] = az;
                sigma[g] = ax * ax + ay * ay + az * az;
            }
        }
    };
    auto stage_and_multiply = [&](const double2 (&src)[4], int c, d4 (&y)[NTO]) { // chunk c of phase 1
#ifdef QCDFT_OCC_ABL_NOCOMPUTE // ablation builds only (tools/occ_ablate.hip)
        for (int r = 0; r < 4; ++r) y[0][r] += src[r].x + src[r].y;
        return;
#endif
        const double *Cc = Cs + (size_t)c * C::CHUNK;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            As[(4 * r + q) * OC_LDA + 2 * li] = src[r].x;
            As[(4 * r + q) * OC_LDA + 2 * li + 1] = src[r].y;
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const double b = As[li * OC_LDA + kq + s];
#pragma unroll
            for (int t = 0; t < NTO; ++t) y[t] = mfma_f64(Cc[(kq + s) * C::LDC + 16 * t + li], b, y[t]);
        }
        __builtin_amdgcn_wave_barrier();
    };
    auto block_dots = [&](const double2 (&g)[3][4], int J, const d4 (&y)[NTO], double (&s1)[4], double (&s2)[4], double (&s3)[4]) {
#ifdef QCDFT_OCC_ABL_NOCOMPUTE
        for (int r = 0; r < 4; ++r) { s1[r] += g[0][r].x + g[0][r].y + y[0][r]; s2[r] += g[1][r].x + g[1][r].y; s3[r] += g[2][r].x + g[2][r].y; }
        return;
#endif
        const double *Ce = Cs + (size_t)J * C::CHUNK + (2 * li) * C::LDC + q, *Co = Ce + C::LDC;
        d4 xe = (d4){0.0, 0.0, 0.0, 0.0}, xo = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int t = 0; t < NTO; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                xe = mfma_f64(y[t][r], Ce[16 * t + 4 * r], xe);
                xo = mfma_f64(y[t][r], Co[16 * t + 4 * r], xo);
            }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            s1[r] += xe[r] * g[0][r].x + xo[r] * g[0][r].y;
            s2[r] += xe[r] * g[1][r].x + xo[r] * g[1][r].y;
            s3[r] += xe[r] * g[2][r].x + xo[r] * g[2][r].y;
        }
    };

    int it = 0, flushed = 0;
    issue_ao(av, blockIdx.x, wave);
    if (GRAD) issue_g(gv, blockIdx.x, wave);
    // all of C, zero-padded to [32 nch][NOP], once per (persistent) workgroup, straight from the caller's (nao, nocc)
    // array -- under the first tile's loads
    for (int e = tid; e < OC_KC * nch * C::NOP; e += 256) {
        const int nu = e / C::NOP, o = e % C::NOP;
        Cs[nu * C::LDC + o] = (nu < nao && o < nocc) ? cocc[(size_t)nu * nocc + o] : 0.0;
    }
    __syncthreads();
    for (long tile = blockIdx.x; tile < ntile; tile += gridDim.x, ++it) {
        const long next = tile + gridDim.x;
        d4 y[NTO];
#pragma unroll
        for (int t = 0; t < NTO; ++t) y[t] = (d4){0.0, 0.0, 0.0, 0.0};
        // ------------------------------------------------ phase 1: this wave's k-chunks of Y^T = C^T . AO^T
        if (mine) stage_and_multiply(av, wave, y);
        else __builtin_amdgcn_s_waitcnt(0); // keeps the issue order of a wave without a chunk like the others'
        issue_ao(av, next, wave);
        if (MULTI) {
            for (int c = wave + 4; c < nch; c += 4) { // nao > 128: further chunks, loaded in place
                double2 ax[4];
                issue_ao(ax, tile, c);
                stage_and_multiply(ax, c, y);
            }
        }
        // partial Y^T of this wave, registers as they are: slot (t, lane) holds the d4
#ifndef QCDFT_OCC_ABL_NOEXCHANGE
#pragma unroll
        for (int t = 0; t < NTO; ++t) *reinterpret_cast<d4 *>(As + (t * 64 + lane) * 4) = y[t];
#endif
        lds_barrier(); // B1: partials (and the previous tile's gradient partials) are in LDS
        if (GRAD && wave == 0 && it > 0) finish_rows(it - 1, (it - 1) & 1);
#ifndef QCDFT_OCC_ABL_NOEXCHANGE
#pragma unroll
        for (int t = 0; t < NTO; ++t) {
            const d4 a = *reinterpret_cast<const d4 *>(region + 0 * YW + (t * 64 + lane) * 4);
            const d4 b = *reinterpret_cast<const d4 *>(region + 1 * YW + (t * 64 + lane) * 4);
            const d4 c = *reinterpret_cast<const d4 *>(region + 2 * YW + (t * 64 + lane) * 4);
            const d4 d = *reinterpret_cast<const d4 *>(region + 3 * YW + (t * 64 + lane) * 4);
            y[t] = (a + b) + (c + d); // the same order in every wave: all four hold bitwise the same Y
        }
#endif
        lds_barrier(); // B2: everyone has read; the regions may take the next AO chunks
        if (it - flushed == OC_OT) { // tiles [flushed, it) are complete in the ring (their gradients since B1 .. B2)
            flush(flushed, it);
            flushed = it;
        }
        if (wave == 1) { // rho of the 16 rows
            double loc = 0.0;
#pragma unroll
            for (int t = 0; t < NTO; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) loc += y[t][r] * y[t][r];
            loc += __shfl_xor(loc, 16, 64);
            loc += __shfl_xor(loc, 32, 64);
            if (lane < 16) Ob[((it % OC_OR) * 16 + lane) * 4] = loc;
        }
        // ------------------------------------------------ phase 2: this wave's column blocks of X = Y . C^T, row dots
        if (GRAD) {
            double s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0}, s3[4] = {0, 0, 0, 0};
            if (mine) block_dots(gv, wave, y, s1, s2, s3);
            else __builtin_amdgcn_s_waitcnt(0);
            issue_g(gv, next, wave);
            if (MULTI) {
                for (int J = wave + 4; J < nch; J += 4) {
                    double2 ge[3][4];
                    issue_g(ge, tile, J);
                    block_dots(ge, J, y, s1, s2, s3);
                }
            }
            double *gp = Gs + (((it & 1) * 4 + wave) * 16) * 3;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
#ifdef QCDFT_OCC_ABL_NOREDUCE
                const double a = s1[r], b = s2[r], c = s3[r];
#else
                const double a = row16_sum(s1[r]), b = row16_sum(s2[r]), c = row16_sum(s3[r]);
#endif
                if (li == 0) {
                    gp[(q + 4 * r) * 3 + 0] = a;
                    gp[(q + 4 * r) * 3 + 1] = b;
                    gp[(q + 4 * r) * 3 + 2] = c;
                }
            }
        }
    }
    lds_barrier();
    if (GRAD && wave == 0 && it > 0) finish_rows(it - 1, (it - 1) & 1);
    lds_barrier();
    flush(flushed, it);
}

// bytes of dynamic LDS of k_rho_occ_rs
inline size_t occ_rs_lds_bytes(int nto, int nch)
{
    const size_t yw = (size_t)std::max(nto * 256, 16 * OC_LDA);
    return sizeof(double) * ((size_t)nch * OC_KC * (16 * nto + 1) + 4 * yw + 2 * 4 * 16 * 3 + (size_t)OC_OR * 16 * 4);
}

} // namespace qcdft
