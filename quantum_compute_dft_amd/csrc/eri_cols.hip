// Electron-repulsion integrals (ab|cd) for ONE ket shell pair (c, d) and ALL bra shell pairs a >= b, on the device:
// the columns of the ERI matrix that the pivoted Cholesky factorisation asks for (cholesky.py), written straight
// into HBM where the factorisation's algebra runs.  Device counterpart of integrals.c::qc_eri_cols2 (same
// McMurchie-Davidson formulation, same Boys function, same screening; the reference takes these integrals from
// libcint, `mol.intor('int2e')` at grid.py:65).  s, p, d and f shells.
//
// One workgroup (256 threads) per bra shell pair; launches are grouped by the angular momenta (la, lb) of the bra,
// so every workgroup of a launch has the same shape and the same LDS layout:
//   per ket primitive pair   Hermite coefficients E^cd (three threads, one per Cartesian direction), then the sparse
//                            products wk[kc][e] = (-1)^(tau+nu+phi) E_x E_y E_z of every ket component pair
//   per bra primitive pair   E^ab and wb[kb][e] likewise (prefactor folded in); the Hermite Coulomb integrals
//                            R_tuv of the primitive quartet by the downward recurrence in the auxiliary index
//                            (two flat (L+1)^3 tables, one barrier per level, all entries of a level in parallel);
//                            then every thread adds  sum_e sum_f wb[kb][e] wk[kc][f] R[o_e + o_f]  to the
//                            Cartesian integrals it owns (NACC accumulators in registers)
//   at the end               the Cartesian block goes to LDS, is rotated to real solid harmonics in place (one
//                            index at a time: a thread owns a column of the rotated index) and scattered into
//                            out[(k, l)][i][j], i >= j only (every (ij|kl) matrix is symmetric: the consumer mirrors).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <new>
#include <vector>

#include "../../include/dft_solver.h"

namespace {

constexpr int EC_T = 256;
constexpr int EC_MAXC = 10;   // Cartesian components of an f shell
constexpr int EC_ED = 4 * 4 * 7; // E[i][j][t], i, j <= 3, t <= 6

__constant__ int c_cx[4][EC_MAXC], c_cy[4][EC_MAXC], c_cz[4][EC_MAXC];
__constant__ double c_sph[4][7][EC_MAXC];

struct EriDev {
    int nshell = 0, nao = 0, npairs = 0;
    double *xyz = nullptr, *ex = nullptr, *cf = nullptr, *qmax = nullptr;
    int *ls = nullptr, *nprim = nullptr, *off = nullptr, *ao0 = nullptr, *pA = nullptr, *pB = nullptr;
    int *cls_pairs = nullptr;        // pair indices grouped by (la, lb) class
    int cls_off[17] = {0};           // class (la*4 + lb) -> [cls_off[c], cls_off[c+1])
    std::vector<int> h_ls;
    hipStream_t stream = nullptr;
    // the (la, lb) classes of one call run side by side: a launch lasts as long as its slowest workgroup (a pair of
    // contracted s shells: up to 36 x 36 primitive quartets in sequence), and the classes' slowest ones overlap this way
    hipStream_t side[4] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t fork = nullptr, join[4] = {nullptr, nullptr, nullptr, nullptr};
    char err[256] = {0};
};

__device__ __forceinline__ int ncart(int l) { return (l + 1) * (l + 2) / 2; }

// F_0..F_n(x): integrals.c::boys, term for term
__device__ void boys(int n, double x, double *F)
{
    if (x < 1e-13) {
        for (int m = 0; m <= n; ++m) F[m] = 1.0 / (2 * m + 1);
        return;
    }
    if (x > 40.0) {
        F[0] = 0.5 * sqrt(M_PI / x);
        const double ex = exp(-x);
        for (int m = 0; m < n; ++m) F[m + 1] = ((2 * m + 1) * F[m] - ex) / (2.0 * x);
        return;
    }
    const double ex = exp(-x);
    double term = 1.0 / (2 * n + 1), sum = term;
    for (int k = 1; k < 400; ++k) {
        term *= 2.0 * x / (2 * n + 2 * k + 1);
        sum += term;
        if (term < 1e-17 * sum) break;
    }
    F[n] = ex * sum;
    for (int m = n; m > 0; --m) F[m - 1] = (2.0 * x * F[m] + ex) / (2 * m - 1);
}

// E[i][j][t] of one Cartesian direction (integrals.c::hermite_E), i <= la, j <= lb; E is [4][4][7]
__device__ void hermite_E(int la, int lb, double a, double b, double XAB, double *E)
{
    const double p = a + b, mu = a * b / p, XPA = -b / p * XAB, XPB = a / p * XAB;
    for (int i = 0; i < EC_ED; ++i) E[i] = 0.0;
    auto at = [&](int i, int j, int t) -> double & { return E[(i * 4 + j) * 7 + t]; };
    auto get = [&](int i, int j, int t) -> double { return (t < 0 || t > i + j) ? 0.0 : E[(i * 4 + j) * 7 + t]; };
    at(0, 0, 0) = exp(-mu * XAB * XAB);
    for (int i = 0; i <= la; ++i) {
        if (i > 0)
            for (int t = 0; t <= i; ++t) at(i, 0, t) = XPA * get(i - 1, 0, t) + get(i - 1, 0, t - 1) / (2 * p) + (t + 1) * get(i - 1, 0, t + 1);
        for (int j = 1; j <= lb; ++j)
            for (int t = 0; t <= i + j; ++t) at(i, j, t) = XPB * get(i, j - 1, t) + get(i, j - 1, t - 1) / (2 * p) + (t + 1) * get(i, j - 1, t + 1);
    }
}

// LDS layout (doubles unless said otherwise), sizes by the launch's class.  A workgroup works as 256 / TEAM teams of
// TEAM threads: every team takes its own bra primitive pairs (round-robin) and has its own
//   Ra, Rb : RD^3 each (RD = L + 1)      wb : nab_c * MT      Eab : 3 * 112      F : 16
// while the ket side is shared:  wk : ncd_c * MT,  Ecd : 3 * 112,  lists ob, ok (unsigned short), counts nb, nk (int).
// TEAM = 256: one team, every step between workgroup barriers (the classes with d or f functions on the bra: many
// Cartesian components, one primitive pair).  TEAM = 64 (s and p on the bra: few components, up to 36 primitive pairs
// per shell pair, and (ss|ss)-like quartets walk up to 36 x 36 primitive quartets): four waves, each on its own primitive
// pairs with nothing but wave-level ordering between its steps; their accumulators are added in a fixed order at the end.
// The Cartesian block (nab_c * ncd_c doubles per team) aliases the front of the allocation after the primitive loops.
template <int NACC, int TEAM>
__global__ __launch_bounds__(EC_T) void k_eri_cols(int nao, const double *__restrict__ xyz, const int *__restrict__ ls,
                                                   const int *__restrict__ nprim, const int *__restrict__ off,
                                                   const int *__restrict__ ao0, const double *__restrict__ ex,
                                                   const double *__restrict__ cf, const int *__restrict__ pA,
                                                   const int *__restrict__ pB, const double *__restrict__ qmax,
                                                   const int *__restrict__ pairs, int C, int D, int kcd, int swap,
                                                   double screen, int mt, double *__restrict__ out)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    constexpr int NTEAM = EC_T / TEAM;
    const int tid = threadIdx.x, tl = tid % TEAM;
    const int team = __builtin_amdgcn_readfirstlane(tid / TEAM);
    auto team_sync = [&]() {
        if (TEAM == EC_T) __syncthreads();
        else __builtin_amdgcn_wave_barrier(); // one wave: its LDS operations complete in program order
    };
    const int kab = pairs[blockIdx.x];
    if (qmax[kab] * qmax[kcd] < screen) return; // Schwarz: the block stays zero (the caller cleared `out`)
    const int A = pA[kab], B = pB[kab];
    const int la = ls[A], lb = ls[B], lc = ls[C], ld = ls[D];
    const int nca = ncart(la), ncb = ncart(lb), ncc = ncart(lc), ncd = ncart(ld);
    const int nab = nca * ncb, ncdc = ncc * ncd, nout = nab * ncdc;
    const int Lab = la + lb, Lcd = lc + ld, L = Lab + Lcd, RD = L + 1, RD3 = RD * RD * RD;

    const int TP = 2 * RD3 + nab * mt + 3 * EC_ED + 16; // doubles per team
    double *Ra = lds + (size_t)team * TP, *Rb = Ra + RD3, *wb = Rb + RD3, *Eab = wb + nab * mt, *F = Eab + 3 * EC_ED;
    double *wk = lds + (size_t)NTEAM * TP, *Ecd = wk + ncdc * mt;
    unsigned short *ob = reinterpret_cast<unsigned short *>(Ecd + 3 * EC_ED), *ok = ob + nab * mt;
    int *nb = reinterpret_cast<int *>(ok + ncdc * mt + ((nab * mt + ncdc * mt) & 1)), *nk = nb + nab;

    // sparse Hermite lists of every component pair: offsets o = (t RD + u) RD + v, t <= ax+bx, u <= ay+by, v <= az+bz
    for (int k = tid; k < nab; k += EC_T) {
        const int ia = k / ncb, ib = k - ia * ncb;
        int n = 0;
        for (int t = 0; t <= c_cx[la][ia] + c_cx[lb][ib]; ++t)
            for (int u = 0; u <= c_cy[la][ia] + c_cy[lb][ib]; ++u)
                for (int v = 0; v <= c_cz[la][ia] + c_cz[lb][ib]; ++v) ob[k * mt + n++] = (unsigned short)((t * RD + u) * RD + v);
        nb[k] = n;
    }
    for (int k = tid; k < ncdc; k += EC_T) {
        const int ic = k / ncd, id = k - ic * ncd;
        int n = 0;
        for (int t = 0; t <= c_cx[lc][ic] + c_cx[ld][id]; ++t)
            for (int u = 0; u <= c_cy[lc][ic] + c_cy[ld][id]; ++u)
                for (int v = 0; v <= c_cz[lc][ic] + c_cz[ld][id]; ++v) ok[k * mt + n++] = (unsigned short)((t * RD + u) * RD + v);
        nk[k] = n;
    }

    double acc[NACC];
#pragma unroll
    for (int o = 0; o < NACC; ++o) acc[o] = 0.0;

    const double *RA = xyz + 3 * A, *RB = xyz + 3 * B, *RC = xyz + 3 * C, *RDc = xyz + 3 * D;
    const double Rab2 = (RA[0] - RB[0]) * (RA[0] - RB[0]) + (RA[1] - RB[1]) * (RA[1] - RB[1]) + (RA[2] - RB[2]) * (RA[2] - RB[2]);
    const double Rcd2 = (RC[0] - RDc[0]) * (RC[0] - RDc[0]) + (RC[1] - RDc[1]) * (RC[1] - RDc[1]) + (RC[2] - RDc[2]) * (RC[2] - RDc[2]);
    const double two_pi_52 = 34.986836655249725; // 2 pi^(5/2)
    const int npa = nprim[A], npb = nprim[B];

    for (int pc = 0; pc < nprim[C]; ++pc)
        for (int pd = 0; pd < nprim[D]; ++pd) {
            const double ec = ex[off[C] + pc], ed = ex[off[D] + pd], ccd = cf[off[C] + pc] * cf[off[D] + pd];
            if (fabs(ccd) * exp(-ec * ed / (ec + ed) * Rcd2) < 1e-18) continue; // the host drops these primitive pairs too (uniform)
            const double q = ec + ed;
            const double Q[3] = {(ec * RC[0] + ed * RDc[0]) / q, (ec * RC[1] + ed * RDc[1]) / q, (ec * RC[2] + ed * RDc[2]) / q};
            __syncthreads(); // every team is done with the previous wk / Ecd
            if (tid < 3) hermite_E(lc, ld, ec, ed, RC[tid] - RDc[tid], Ecd + tid * EC_ED);
            __syncthreads();
            for (int k = tid; k < ncdc; k += EC_T) {
                const int ic = k / ncd, id = k - ic * ncd;
                const int x1 = c_cx[lc][ic], x2 = c_cx[ld][id], y1 = c_cy[lc][ic], y2 = c_cy[ld][id], z1 = c_cz[lc][ic], z2 = c_cz[ld][id];
                int n = 0;
                for (int t = 0; t <= x1 + x2; ++t) {
                    const double e1 = Ecd[0 * EC_ED + (x1 * 4 + x2) * 7 + t];
                    for (int u = 0; u <= y1 + y2; ++u) {
                        const double e2 = e1 * Ecd[1 * EC_ED + (y1 * 4 + y2) * 7 + u];
                        for (int v = 0; v <= z1 + z2; ++v) {
                            const double e3 = e2 * Ecd[2 * EC_ED + (z1 * 4 + z2) * 7 + v];
                            wk[k * mt + n++] = ((t + u + v) & 1) ? -e3 : e3;
                        }
                    }
                }
            }
            __syncthreads(); // wk complete for every team
            for (int pab = team; pab < npa * npb; pab += NTEAM) { // this team's bra primitive pairs
                const int pa = pab / npb, pb = pab - pa * npb;
                const double ea = ex[off[A] + pa], eb = ex[off[B] + pb], cab = cf[off[A] + pa] * cf[off[B] + pb];
                if (fabs(cab) * exp(-ea * eb / (ea + eb) * Rab2) < 1e-18) continue; // uniform within the team
                const double p = ea + eb;
                const double P[3] = {(ea * RA[0] + eb * RB[0]) / p, (ea * RA[1] + eb * RB[1]) / p, (ea * RA[2] + eb * RB[2]) / p};
                const double alpha = p * q / (p + q);
                const double PQ[3] = {P[0] - Q[0], P[1] - Q[1], P[2] - Q[2]};
                const double pref = two_pi_52 / (p * q * sqrt(p + q)) * cab * ccd;
                team_sync(); // the team's previous contraction is over: wb, Eab, R tables are free
                if (tl < 3) hermite_E(la, lb, ea, eb, RA[tl] - RB[tl], Eab + tl * EC_ED);
                if (tl == (TEAM == EC_T ? 64 : 3)) { // F_n scaled to R^n_000 = (-2 alpha)^n F_n (the host's order of operations)
                    boys(L, alpha * (PQ[0] * PQ[0] + PQ[1] * PQ[1] + PQ[2] * PQ[2]), F);
                    double f = 1.0;
                    for (int n = 0; n <= L; ++n) { F[n] *= f; f *= -2.0 * alpha; }
                }
                team_sync();
                for (int k = tl; k < nab; k += TEAM) {
                    const int ia = k / ncb, ib = k - ia * ncb;
                    const int x1 = c_cx[la][ia], x2 = c_cx[lb][ib], y1 = c_cy[la][ia], y2 = c_cy[lb][ib], z1 = c_cz[la][ia], z2 = c_cz[lb][ib];
                    int n = 0;
                    for (int t = 0; t <= x1 + x2; ++t) {
                        const double e1 = pref * Eab[0 * EC_ED + (x1 * 4 + x2) * 7 + t];
                        for (int u = 0; u <= y1 + y2; ++u) {
                            const double e2 = e1 * Eab[1 * EC_ED + (y1 * 4 + y2) * 7 + u];
                            for (int v = 0; v <= z1 + z2; ++v) wb[k * mt + n++] = e2 * Eab[2 * EC_ED + (z1 * 4 + z2) * 7 + v];
                        }
                    }
                }
                // R^n_tuv, n = L .. 0 (integrals.c::hermite_R rearranged by auxiliary level): level n holds the orders
                // t+u+v <= L-n and needs level n+1 only
                double *cur = Ra, *nxt = Rb;
                if (tl == 0) cur[0] = F[L];
                for (int n = L - 1; n >= 0; --n) {
                    team_sync();
                    const int smax = L - n, side = smax + 1;
                    for (int e = tl; e < side * side * side; e += TEAM) {
                        const int t = e / (side * side), u = (e / side) % side, v = e % side;
                        const int s = t + u + v;
                        if (s > smax) continue;
                        double val;
                        if (s == 0) {
                            val = F[n];
                        } else if (t > 0) {
                            val = PQ[0] * cur[((t - 1) * RD + u) * RD + v];
                            if (t > 1) val += (t - 1) * cur[((t - 2) * RD + u) * RD + v];
                        } else if (u > 0) {
                            val = PQ[1] * cur[(t * RD + u - 1) * RD + v];
                            if (u > 1) val += (u - 1) * cur[(t * RD + u - 2) * RD + v];
                        } else {
                            val = PQ[2] * cur[(t * RD + u) * RD + v - 1];
                            if (v > 1) val += (v - 1) * cur[(t * RD + u) * RD + v - 2];
                        }
                        nxt[(t * RD + u) * RD + v] = val;
                    }
                    double *sw = cur; cur = nxt; nxt = sw;
                }
                team_sync(); // `cur` = R^0, wb complete
#pragma unroll
                for (int o = 0; o < NACC; ++o) {
                    int e = tl + o * TEAM;
                    // opaque to the optimiser: otherwise the index arithmetic of all NACC outputs (kb, kc, four pointers, two
                    // counts each) is hoisted out of the primitive loops and kept in registers -- 190 of them besides the
                    // accumulators at NACC = 16, scratch at NACC = 40
                    if (NACC >= 40) asm volatile("" : "+v"(e));
                    if (e < nout) {
                        const int kb = e / ncdc, kc = e - kb * ncdc;
                        const int n1 = nb[kb], n2 = nk[kc];
                        const double *w1 = wb + kb * mt, *w2 = wk + kc * mt;
                        const unsigned short *o1 = ob + kb * mt, *o2 = ok + kc * mt;
                        double s = 0.0;
                        for (int i = 0; i < n1; ++i) {
                            const double *Rm = cur + o1[i];
                            double g = 0.0;
                            for (int j = 0; j < n2; ++j) g += w2[j] * Rm[o2[j]];
                            s += w1[i] * g;
                        }
                        acc[o] += s;
                    }
                }
            }
        }

    // the teams' Cartesian partial sums -> LDS, added in team order into team 0's copy: [ca][cb][cc][cd]
    __syncthreads();
    double *cart = lds;
#pragma unroll
    for (int o = 0; o < NACC; ++o) {
        const int e = tl + o * TEAM;
        if (e < nout) cart[(size_t)team * nout + e] = acc[o];
    }
    __syncthreads();
    if (NTEAM > 1) {
        for (int e = tid; e < nout; e += EC_T) {
            double v = cart[e];
            for (int w = 1; w < NTEAM; ++w) v += cart[(size_t)w * nout + e];
            cart[e] = v;
        }
        __syncthreads();
    }
    // rotated in place index by index, scattered
    const int nsa = 2 * la + 1, nsb = 2 * lb + 1, nsc = 2 * lc + 1, nsd = 2 * ld + 1;
    // One index at a time, in place: a thread owns one column of the index being rotated (reads its <= 10 Cartesian
    // entries, writes its <= 7 spherical ones over them); the strides stay the Cartesian ones.
    auto rotate_column = [&](int l, int nc, int ns, double *base, int stride) {
        double in[EC_MAXC];
#pragma unroll
        for (int c = 0; c < EC_MAXC; ++c) in[c] = c < nc ? base[c * stride] : 0.0;
        for (int m = 0; m < ns; ++m) {
            double s = 0.0;
#pragma unroll
            for (int c = 0; c < EC_MAXC; ++c) s += c_sph[l][m][c] * in[c]; // rows of c_sph are zero past nc
            base[m * stride] = s;
        }
    };
    const int sb = ncc * ncd, sa = ncb * sb;
    for (int i = tid; i < sa; i += EC_T) rotate_column(la, nca, nsa, cart + i, sa);                       // a: columns (cb, cc, cd)
    __syncthreads();
    for (int col = tid; col < nsa * sb; col += EC_T) {                                                     // b: columns (a, cc, cd)
        const int a = col / sb, i = col - a * sb;
        rotate_column(lb, ncb, nsb, cart + (size_t)a * sa + i, sb);
    }
    __syncthreads();
    for (int col = tid; col < nsa * nsb * ncd; col += EC_T) {                                              // c: columns (a, b, cd)
        const int ab = col / ncd, i = col - ab * ncd, a = ab / nsb, b = ab - a * nsb;
        rotate_column(lc, ncc, nsc, cart + (size_t)a * sa + (size_t)b * sb + i, ncd);
    }
    __syncthreads();
    // d, and the scatter: element (a, b, c3, c4) of the stored (C, D) order
    const size_t n2 = (size_t)nao * nao;
    for (int col = tid; col < nsa * nsb * nsc; col += EC_T) {
        const int ab = col / nsc, c3 = col - ab * nsc, a = ab / nsb, b = ab - a * nsb;
        const double *base = cart + (size_t)a * sa + (size_t)b * sb + c3 * ncd;
        double in[EC_MAXC];
#pragma unroll
        for (int c = 0; c < EC_MAXC; ++c) in[c] = c < ncd ? base[c] : 0.0;
        const size_t i = ao0[A] + a, j = ao0[B] + b;
        for (int c4 = 0; c4 < nsd; ++c4) {
            double s = 0.0;
#pragma unroll
            for (int c = 0; c < EC_MAXC; ++c) s += c_sph[ld][c4][c] * in[c];
            // requested (k over the caller's first shell, l over its second); the pair is stored as (max, min)
            const int mat = swap ? c4 * nsc + c3 : c3 * nsd + c4;
            out[(size_t)mat * n2 + i * nao + j] = s;
        }
    }
}

void fill_tables()
{
    int cx[4][EC_MAXC] = {{0}}, cy[4][EC_MAXC] = {{0}}, cz[4][EC_MAXC] = {{0}};
    for (int l = 0; l < 4; ++l) {
        int n = 0;
        for (int lx = l; lx >= 0; --lx)
            for (int ly = l - lx; ly >= 0; --ly) { cx[l][n] = lx; cy[l][n] = ly; cz[l][n] = l - lx - ly; ++n; }
    }
    double T[4][7][EC_MAXC];
    memset(T, 0, sizeof T);
    T[0][0][0] = 0.282094791773878143;
    for (int i = 0; i < 3; ++i) T[1][i][i] = 0.488602511902919921;
    { // l = 2: xx xy xz yy yz zz (integrals.c::sph_matrix)
        const double c = 1.092548430592079070, d = 0.315391565252520002, e = 0.546274215296039535;
        T[2][0][1] = c; T[2][1][4] = c;
        T[2][2][0] = -d; T[2][2][3] = -d; T[2][2][5] = 2 * d;
        T[2][3][2] = c;
        T[2][4][0] = e; T[2][4][3] = -e;
    }
    { // l = 3: xxx xxy xxz xyy xyz xzz yyy yyz yzz zzz
        const double f3 = 0.590043589926643510, f2 = 2.890611442640554055, f1 = 0.457045799464465739,
                     f0 = 0.373176332590115391, f2b = 1.445305721320277020;
        T[3][0][1] = 3 * f3; T[3][0][6] = -f3;
        T[3][1][4] = f2;
        T[3][2][8] = 4 * f1; T[3][2][1] = -f1; T[3][2][6] = -f1;
        T[3][3][9] = 2 * f0; T[3][3][2] = -3 * f0; T[3][3][7] = -3 * f0;
        T[3][4][5] = 4 * f1; T[3][4][0] = -f1; T[3][4][3] = -f1;
        T[3][5][2] = f2b; T[3][5][7] = -f2b;
        T[3][6][0] = f3; T[3][6][3] = -3 * f3;
    }
    (void)hipMemcpyToSymbol(HIP_SYMBOL(c_cx), cx, sizeof cx);
    (void)hipMemcpyToSymbol(HIP_SYMBOL(c_cy), cy, sizeof cy);
    (void)hipMemcpyToSymbol(HIP_SYMBOL(c_cz), cz, sizeof cz);
    (void)hipMemcpyToSymbol(HIP_SYMBOL(c_sph), T, sizeof T);
}

int max_terms(int la, int lb)
{
    int best = 1;
    for (int ax = 0; ax <= la; ++ax)
        for (int ay = 0; ay <= la - ax; ++ay)
            for (int bx = 0; bx <= lb; ++bx)
                for (int by = 0; by <= lb - bx; ++by) {
                    const int az = la - ax - ay, bz = lb - bx - by;
                    best = std::max(best, (ax + bx + 1) * (ay + by + 1) * (az + bz + 1));
                }
    return best;
}

template <class T> bool upload(T *&dst, const T *src, size_t n)
{
    if (hipMalloc((void **)&dst, sizeof(T) * std::max<size_t>(n, 1)) != hipSuccess) return false;
    return hipMemcpy(dst, src, sizeof(T) * n, hipMemcpyHostToDevice) == hipSuccess;
}

} // namespace

extern "C" {

void *DFT_EriColumnsOpen(int nshell, const double *xyz, const int *ls, const int *nprim, const int *off, const int *ao0,
                         const double *ex, const double *cf, int nao, int nprim_total, const double *qmax_pairs)
{
    for (int s = 0; s < nshell; ++s)
        if (ls[s] < 0 || ls[s] > 3) return nullptr;
    EriDev *c = new (std::nothrow) EriDev();
    if (!c) return nullptr;
    c->nshell = nshell; c->nao = nao; c->npairs = nshell * (nshell + 1) / 2;
    c->h_ls.assign(ls, ls + nshell);
    std::vector<int> pA(c->npairs), pB(c->npairs);
    for (int A = 0, k = 0; A < nshell; ++A)
        for (int B = 0; B <= A; ++B, ++k) { pA[k] = A; pB[k] = B; }
    std::vector<int> grouped;
    grouped.reserve(c->npairs);
    for (int cls = 0; cls < 16; ++cls) {
        c->cls_off[cls] = (int)grouped.size();
        for (int k = 0; k < c->npairs; ++k)
            if (ls[pA[k]] * 4 + ls[pB[k]] == cls) grouped.push_back(k);
        // most primitive pairs first: a workgroup's time goes with nprim(a) nprim(b), and the few contracted-contracted
        // pairs of a class (36 primitive pairs against 1) would otherwise start late and finish alone
        std::stable_sort(grouped.begin() + c->cls_off[cls], grouped.end(),
                         [&](int x, int y) { return nprim[pA[x]] * nprim[pB[x]] > nprim[pA[y]] * nprim[pB[y]]; });
    }
    c->cls_off[16] = (int)grouped.size();
    fill_tables();
    const bool ok = upload(c->xyz, xyz, 3 * (size_t)nshell) && upload(c->ex, ex, (size_t)nprim_total) && upload(c->cf, cf, (size_t)nprim_total) &&
                    upload(c->qmax, qmax_pairs, (size_t)c->npairs) && upload(c->ls, ls, (size_t)nshell) && upload(c->nprim, nprim, (size_t)nshell) &&
                    upload(c->off, off, (size_t)nshell) && upload(c->ao0, ao0, (size_t)nshell) && upload(c->pA, pA.data(), pA.size()) &&
                    upload(c->pB, pB.data(), pB.size()) && upload(c->cls_pairs, grouped.data(), grouped.size());
    bool ok2 = ok && hipEventCreateWithFlags(&c->fork, hipEventDisableTiming) == hipSuccess;
    for (int i = 0; i < 4 && ok2; ++i)
        ok2 = hipStreamCreateWithFlags(&c->side[i], hipStreamNonBlocking) == hipSuccess &&
              hipEventCreateWithFlags(&c->join[i], hipEventDisableTiming) == hipSuccess;
    if (!ok2) {
        (void)hipGetLastError();
        DFT_EriColumnsClose(c);
        return nullptr;
    }
    return c;
}

void DFT_EriColumnsClose(void *h)
{
    EriDev *c = (EriDev *)h;
    if (!c) return;
    void *bufs[] = {c->xyz, c->ex, c->cf, c->qmax, c->ls, c->nprim, c->off, c->ao0, c->pA, c->pB, c->cls_pairs};
    for (void *b : bufs)
        if (b) (void)hipFree(b);
    for (int i = 0; i < 4; ++i) {
        if (c->side[i]) { (void)hipStreamSynchronize(c->side[i]); (void)hipStreamDestroy(c->side[i]); }
        if (c->join[i]) (void)hipEventDestroy(c->join[i]);
    }
    if (c->fork) (void)hipEventDestroy(c->fork);
    delete c;
}

int DFT_EriColumnsSetStream(void *h, unsigned long long hip_stream)
{
    EriDev *c = (EriDev *)h;
    if (!c) return -1;
    c->stream = (hipStream_t)hip_stream;
    return 0;
}

const char *DFT_EriColumnsLastError(void *h) { return h ? ((EriDev *)h)->err : "null handle"; }

} // extern "C"

// The launches of one ket shell pair into `out` (already cleared), dealt to the side streams from launch counter `nlaunch` on.
static int launch_pair(EriDev *c, int C, int D, double screen, double *out, int &nlaunch)
{
    const int swap = C < D, Cs = swap ? D : C, Ds = swap ? C : D; // stored pair (max, min)
    const int kcd = Cs * (Cs + 1) / 2 + Ds;
    const int lc = c->h_ls[Cs], ld = c->h_ls[Ds];
    const int ncdc = (lc + 1) * (lc + 2) / 2 * ((ld + 1) * (ld + 2) / 2);
    const int mtk = max_terms(lc, ld);
    for (int la = 0; la < 4; ++la)
        for (int lb = 0; lb <= la; ++lb) {
            // shell pairs are stored A >= B by INDEX, so both (la, lb) and (lb, la) classes occur
            for (int flip = 0; flip < (la == lb ? 1 : 2); ++flip) {
                const int l1 = flip ? lb : la, l2 = flip ? la : lb, cls = l1 * 4 + l2;
                const int cnt = c->cls_off[cls + 1] - c->cls_off[cls];
                if (!cnt) continue;
                const int nab = (l1 + 1) * (l1 + 2) / 2 * ((l2 + 1) * (l2 + 2) / 2);
                const int mt = std::max(max_terms(l1, l2), mtk);
                const int L = l1 + l2 + lc + ld, RD = L + 1;
                const size_t nout = (size_t)nab * ncdc;
                const int team = (l1 <= 1 && l2 <= 1) ? 64 : EC_T;   // s / p on the bra: a wave per primitive pair
                const int nteam = EC_T / team;
                const size_t tp = 2 * (size_t)RD * RD * RD + (size_t)nab * mt + 3 * EC_ED + 16;
                const size_t dbl = nteam * tp + (size_t)ncdc * mt + 3 * EC_ED;
                size_t bytes = dbl * 8 + ((size_t)(nab + ncdc) * mt + 1) * 2 + (size_t)(nab + ncdc) * 4 + 16;
                bytes = std::max(bytes, nteam * nout * 8);
                bytes = (bytes + 15) & ~(size_t)15;
                const int nacc = (int)((nout + team - 1) / team);
                const int *pairs = c->cls_pairs + c->cls_off[cls];
#define QC_ERI_LAUNCH(N, TEAMSZ)                                                                                             \
    do {                                                                                                                     \
        auto kern = k_eri_cols<N, TEAMSZ>;                                                                                   \
        static size_t allowed = 48 * 1024;                                                                                   \
        if (bytes > allowed) {                                                                                               \
            if (hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess) { \
                snprintf(c->err, sizeof c->err, "LDS request of %zu bytes refused", bytes);                                  \
                return -1;                                                                                                   \
            }                                                                                                                \
            allowed = bytes;                                                                                                 \
        }                                                                                                                    \
        hipLaunchKernelGGL(kern, dim3((unsigned)cnt), dim3(EC_T), bytes, c->side[nlaunch++ & 3], c->nao, c->xyz, c->ls, c->nprim, c->off, c->ao0, c->ex, \
                           c->cf, c->pA, c->pB, c->qmax, pairs, Cs, Ds, kcd, swap, screen, mt, out);                         \
    } while (0)
                if (team == 64) { // nout <= 9 x 100
                    if (nacc <= 1) QC_ERI_LAUNCH(1, 64);
                    else if (nacc <= 2) QC_ERI_LAUNCH(2, 64);
                    else if (nacc <= 4) QC_ERI_LAUNCH(4, 64);
                    else if (nacc <= 8) QC_ERI_LAUNCH(8, 64);
                    else QC_ERI_LAUNCH(16, 64);
                } else {
                    if (nacc <= 1) QC_ERI_LAUNCH(1, 256);
                    else if (nacc <= 2) QC_ERI_LAUNCH(2, 256);
                    else if (nacc <= 4) QC_ERI_LAUNCH(4, 256);
                    else if (nacc <= 8) QC_ERI_LAUNCH(8, 256);
                    else if (nacc <= 16) QC_ERI_LAUNCH(16, 256);
                    else QC_ERI_LAUNCH(40, 256);
                }
#undef QC_ERI_LAUNCH
            }
        }
    return 0;
}

static int columns_many(EriDev *c, int npairs, const int *Cs, const int *Ds, double screen, double *out, const long long *offsets)
{
    c->err[0] = 0;
    size_t total = 0;
    for (int k = 0; k < npairs; ++k) {
        if (Cs[k] < 0 || Ds[k] < 0 || Cs[k] >= c->nshell || Ds[k] >= c->nshell) return -1;
        const size_t nq = (size_t)(2 * c->h_ls[Cs[k]] + 1) * (2 * c->h_ls[Ds[k]] + 1);
        if (offsets[k] < 0) return -1;
        total = std::max(total, (size_t)offsets[k] + nq * c->nao * c->nao);
    }
    if (hipMemsetAsync(out, 0, sizeof(double) * total, c->stream) != hipSuccess) {
        snprintf(c->err, sizeof c->err, "memset of the column block failed");
        return -1;
    }
    // fork: the side streams start behind the clear (and whatever the caller queued before it on the handle's stream)
    (void)hipEventRecord(c->fork, c->stream);
    for (int i = 0; i < 4; ++i) (void)hipStreamWaitEvent(c->side[i], c->fork, 0);
    int nlaunch = 0;
    for (int k = 0; k < npairs; ++k)
        if (launch_pair(c, Cs[k], Ds[k], screen, out + offsets[k], nlaunch) != 0) return -1;
    for (int i = 0; i < 4; ++i) { // join: the handle's stream continues behind all of them
        (void)hipEventRecord(c->join[i], c->side[i]);
        (void)hipStreamWaitEvent(c->stream, c->join[i], 0);
    }
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        snprintf(c->err, sizeof c->err, "ERI column launch failed: %s", hipGetErrorString(e));
        return -1;
    }
    return 0;
}

extern "C" {

int DFT_EriColumns(void *h, int C, int D, double screen, unsigned long long d_out)
{
    EriDev *c = (EriDev *)h;
    if (!c || !d_out) return -1;
    const long long zero = 0;
    return columns_many(c, 1, &C, &D, screen, (double *)d_out, &zero);
}

// Several ket shell pairs in one call: block k (as DFT_EriColumns lays it out) starts offsets[k] doubles into d_out; the
// blocks must not overlap and [0, max end) is cleared as a whole.  One fork / join around all launches: the pairs' kernels
// run side by side instead of pair after pair.
int DFT_EriColumnsMany(void *h, int npairs, const int *C, const int *D, double screen, unsigned long long d_out, const long long *offsets)
{
    EriDev *c = (EriDev *)h;
    if (!c || !d_out || npairs <= 0 || !C || !D || !offsets) return -1;
    return columns_many(c, npairs, C, D, screen, (double *)d_out, offsets);
}

} // extern "C"
