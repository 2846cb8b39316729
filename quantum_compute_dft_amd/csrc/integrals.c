/*
 * Gaussian integrals for the SCF driver: overlap, kinetic, nuclear attraction and the dense
 * two-electron tensor over contracted real-spherical shells (s, p, d, f).
 *
 * Host-side counterpart of what the reference obtains from PySCF/libcint at grid.py:61-65
 * (mol.intor('int1e_ovlp' | 'int1e_kin' | 'int1e_nuc' | 'int2e')), same conventions: real
 * spherical AOs in the shell order of basis.py, ERI in chemists' notation (ij|kl), full
 * nao^4 storage without symmetry packing (what dft.py:166 reshapes to (nao^2, nao^2)).
 *
 * Method: McMurchie-Davidson.  Primitive Cartesian Gaussians x^i y^j z^k exp(-a r^2) are
 * expanded in Hermite Gaussians (E coefficients), Coulomb-type integrals come from the Hermite
 * integrals R_tuv built on the Boys function; contraction coefficients (radial normalisation
 * included, basis.py) are applied per primitive pair and each Cartesian shell block is rotated
 * to real solid harmonics at the end.  OpenMP over shell pairs.  Plain C, no dependencies.
 */
#include <math.h>
#ifdef _OPENMP
#include <omp.h>
#endif
#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif
#include <stdlib.h>
#include <string.h>

#define LMAX 3
#define NCART(l) (((l) + 1) * ((l) + 2) / 2)
#define MAXCART 10
#define HDIM (2 * LMAX + 1)          /* t index range of one shell pair: 0..la+lb */
#define RDIM (4 * LMAX + 1)          /* t index range of R: 0..la+lb+lc+ld */

/* ---- Cartesian component tables and real-solid-harmonic rotation ------------------------- */
static void cart_components(int l, int cx[MAXCART], int cy[MAXCART], int cz[MAXCART])
{
    int n = 0;
    for (int lx = l; lx >= 0; --lx)
        for (int ly = l - lx; ly >= 0; --ly) { cx[n] = lx; cy[n] = ly; cz[n] = l - lx - ly; ++n; }
}

/* T[m][c]: real solid harmonic m of degree l as a combination of Cartesian monomials
 * (orders: p x,y,z; d xy,yz,z2,xz,x2-y2; f m=-3..3), normalisation of ao_kernels.hpp. */
static void sph_matrix(int l, double T[7][MAXCART])
{
    memset(T, 0, sizeof(double) * 7 * MAXCART);
    if (l == 0) { T[0][0] = 0.282094791773878143; return; }
    if (l == 1) { for (int i = 0; i < 3; ++i) T[i][i] = 0.488602511902919921; return; }
    if (l == 2) { /* xx xy xz yy yz zz */
        const double c = 1.092548430592079070, d = 0.315391565252520002, e = 0.546274215296039535;
        T[0][1] = c; T[1][4] = c;
        T[2][0] = -d; T[2][3] = -d; T[2][5] = 2 * d;
        T[3][2] = c;
        T[4][0] = e; T[4][3] = -e;
        return;
    }
    /* l == 3: xxx xxy xxz xyy xyz xzz yyy yyz yzz zzz */
    const double f3 = 0.590043589926643510, f2 = 2.890611442640554055, f1 = 0.457045799464465739,
                 f0 = 0.373176332590115391, f2b = 1.445305721320277020;
    T[0][1] = 3 * f3; T[0][6] = -f3;
    T[1][4] = f2;
    T[2][8] = 4 * f1; T[2][1] = -f1; T[2][6] = -f1;
    T[3][9] = 2 * f0; T[3][2] = -3 * f0; T[3][7] = -3 * f0;
    T[4][5] = 4 * f1; T[4][0] = -f1; T[4][3] = -f1;
    T[5][2] = f2b; T[5][7] = -f2b;
    T[6][0] = f3; T[6][3] = -3 * f3;
}

/* ---- Boys function F_0..F_n(x) -------------------------------------------------------------- */
static void boys(int n, double x, double *F)
{
    if (x < 1e-13) { for (int m = 0; m <= n; ++m) F[m] = 1.0 / (2 * m + 1); return; }
    if (x > 40.0) { /* erf-type asymptote, then stable upward recursion */
        F[0] = 0.5 * sqrt(M_PI / x);
        const double ex = exp(-x);
        for (int m = 0; m < n; ++m) F[m + 1] = ((2 * m + 1) * F[m] - ex) / (2.0 * x);
        return;
    }
    /* series for the highest order, downward recursion for the rest */
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

/* ---- Hermite expansion coefficients E[i][j][t] for one dimension --------------------------- */
static void hermite_E(int la, int lb, double a, double b, double XAB, double E[LMAX + 3][LMAX + 3][2 * LMAX + 5])
{
    const double p = a + b, mu = a * b / p, XPA = -b / p * XAB, XPB = a / p * XAB; /* XAB = A - B */
    memset(E, 0, sizeof(double) * (LMAX + 3) * (LMAX + 3) * (2 * LMAX + 5));
    E[0][0][0] = exp(-mu * XAB * XAB);
    for (int i = 0; i <= la; ++i) {
        if (i > 0)
            for (int t = 0; t <= i; ++t) {
                double v = XPA * E[i - 1][0][t];
                if (t > 0) v += E[i - 1][0][t - 1] / (2 * p);
                v += (t + 1) * E[i - 1][0][t + 1];
                E[i][0][t] = v;
            }
        for (int j = 1; j <= lb; ++j)
            for (int t = 0; t <= i + j; ++t) {
                double v = XPB * E[i][j - 1][t];
                if (t > 0) v += E[i][j - 1][t - 1] / (2 * p);
                v += (t + 1) * E[i][j - 1][t + 1];
                E[i][j][t] = v;
            }
    }
}

/* ---- Hermite Coulomb integrals R[t][u][v] (order 0), t+u+v <= L ----------------------------- */
static void hermite_R(int L, double alpha, const double PQ[3], double R[RDIM][RDIM][RDIM])
{
    static _Thread_local double Rn[RDIM + 1][RDIM][RDIM][RDIM];
    double F[RDIM + 1];
    const double r2 = PQ[0] * PQ[0] + PQ[1] * PQ[1] + PQ[2] * PQ[2];
    boys(L, alpha * r2, F);
    double f = 1.0;
    for (int n = 0; n <= L; ++n) { Rn[n][0][0][0] = f * F[n]; f *= -2.0 * alpha; }
    for (int s = 1; s <= L; ++s)            /* total order t+u+v = s, needs auxiliary n <= L - s */
        for (int n = 0; n <= L - s; ++n)
            for (int t = 0; t <= s; ++t)
                for (int u = 0; u <= s - t; ++u) {
                    const int v = s - t - u;
                    double val;
                    if (t > 0) {
                        val = PQ[0] * Rn[n + 1][t - 1][u][v];
                        if (t > 1) val += (t - 1) * Rn[n + 1][t - 2][u][v];
                    } else if (u > 0) {
                        val = PQ[1] * Rn[n + 1][t][u - 1][v];
                        if (u > 1) val += (u - 1) * Rn[n + 1][t][u - 2][v];
                    } else {
                        val = PQ[2] * Rn[n + 1][t][u][v - 1];
                        if (v > 1) val += (v - 1) * Rn[n + 1][t][u][v - 2];
                    }
                    Rn[n][t][u][v] = val;
                }
    for (int t = 0; t <= L; ++t)
        for (int u = 0; u <= L - t; ++u)
            for (int v = 0; v <= L - t - u; ++v) R[t][u][v] = Rn[0][t][u][v];
}

/* rotate a Cartesian block (na x nb) to spherical on both indices and scatter */
static void put_sph2(int la, int lb, const double *cart, double *out, int ld, int ia, int ib)
{
    double Ta[7][MAXCART], Tb[7][MAXCART];
    sph_matrix(la, Ta);
    sph_matrix(lb, Tb);
    const int nca = NCART(la), ncb = NCART(lb);
    for (int ma = 0; ma < 2 * la + 1; ++ma)
        for (int mb = 0; mb < 2 * lb + 1; ++mb) {
            double s = 0.0;
            for (int ca = 0; ca < nca; ++ca) {
                if (Ta[ma][ca] == 0.0) continue;
                for (int cb = 0; cb < ncb; ++cb) s += Ta[ma][ca] * Tb[mb][cb] * cart[ca * ncb + cb];
            }
            out[(size_t)(ia + ma) * ld + ib + mb] = s;
        }
}

/* ---- one-electron integrals ----------------------------------------------------------------- */
int qc_int1e(int nshell, const double *xyz, const int *ls, const int *nprim, const int *off,
             const int *ao0, const double *ex, const double *cf, int nao, int natm,
             const double *atm_xyz, const double *atm_z, double *S, double *T, double *V)
{
    for (int s = 0; s < nshell; ++s)
        if (ls[s] < 0 || ls[s] > LMAX) return -1;
#pragma omp parallel for schedule(dynamic) collapse(2)
    for (int A = 0; A < nshell; ++A)
        for (int B = 0; B < nshell; ++B) {
            const int la = ls[A], lb = ls[B], nca = NCART(la), ncb = NCART(lb);
            int ax[MAXCART], ay[MAXCART], az[MAXCART], bx[MAXCART], by[MAXCART], bz[MAXCART];
            cart_components(la, ax, ay, az);
            cart_components(lb, bx, by, bz);
            double cs[MAXCART * MAXCART] = {0}, ct[MAXCART * MAXCART] = {0}, cv[MAXCART * MAXCART] = {0};
            const double *RA = xyz + 3 * A, *RB = xyz + 3 * B;
            for (int pa = 0; pa < nprim[A]; ++pa)
                for (int pb = 0; pb < nprim[B]; ++pb) {
                    const double a = ex[off[A] + pa], b = ex[off[B] + pb], p = a + b;
                    const double cc = cf[off[A] + pa] * cf[off[B] + pb];
                    double E[3][LMAX + 3][LMAX + 3][2 * LMAX + 5];
                    for (int d = 0; d < 3; ++d) hermite_E(la, lb + 2, a, b, RA[d] - RB[d], E[d]);
                    const double P[3] = {(a * RA[0] + b * RB[0]) / p, (a * RA[1] + b * RB[1]) / p,
                                         (a * RA[2] + b * RB[2]) / p};
                    const double s0 = pow(M_PI / p, 1.5);
                    for (int ca = 0; ca < nca; ++ca)
                        for (int cb = 0; cb < ncb; ++cb) {
                            const int i[3] = {ax[ca], ay[ca], az[ca]}, j[3] = {bx[cb], by[cb], bz[cb]};
                            double s1[3], k1[3];
                            for (int d = 0; d < 3; ++d) {
                                s1[d] = E[d][i[d]][j[d]][0];
                                /* -1/2 d^2/dx^2 acting on the ket */
                                double k = -2.0 * b * b * E[d][i[d]][j[d] + 2][0] + b * (2 * j[d] + 1) * s1[d];
                                if (j[d] >= 2) k -= 0.5 * j[d] * (j[d] - 1) * E[d][i[d]][j[d] - 2][0];
                                k1[d] = k;
                            }
                            cs[ca * ncb + cb] += cc * s0 * s1[0] * s1[1] * s1[2];
                            ct[ca * ncb + cb] += cc * s0 * (k1[0] * s1[1] * s1[2] + s1[0] * k1[1] * s1[2] + s1[0] * s1[1] * k1[2]);
                        }
                    /* nuclear attraction */
                    const int L = la + lb;
                    for (int c = 0; c < natm; ++c) {
                        static _Thread_local double R[RDIM][RDIM][RDIM];
                        const double PC[3] = {P[0] - atm_xyz[3 * c], P[1] - atm_xyz[3 * c + 1], P[2] - atm_xyz[3 * c + 2]};
                        hermite_R(L, p, PC, R);
                        const double pref = -atm_z[c] * 2.0 * M_PI / p * cc;
                        for (int ca = 0; ca < nca; ++ca)
                            for (int cb = 0; cb < ncb; ++cb) {
                                double v = 0.0;
                                for (int t = 0; t <= ax[ca] + bx[cb]; ++t)
                                    for (int u = 0; u <= ay[ca] + by[cb]; ++u)
                                        for (int w = 0; w <= az[ca] + bz[cb]; ++w)
                                            v += E[0][ax[ca]][bx[cb]][t] * E[1][ay[ca]][by[cb]][u] *
                                                 E[2][az[ca]][bz[cb]][w] * R[t][u][w];
                                cv[ca * ncb + cb] += pref * v;
                            }
                    }
                }
            put_sph2(la, lb, cs, S, nao, ao0[A], ao0[B]);
            put_sph2(la, lb, ct, T, nao, ao0[A], ao0[B]);
            put_sph2(la, lb, cv, V, nao, ao0[A], ao0[B]);
        }
    return 0;
}

/* ---- two-electron integrals ------------------------------------------------------------------ */
typedef struct {
    double p, P[3], cc;
    double E[3][LMAX + 1][LMAX + 1][HDIM]; /* E[d][i][j][t] */
} PrimPair;

static void make_pair(int la, int lb, double a, double b, const double *RA, const double *RB, double cc, PrimPair *pp)
{
    double E[LMAX + 3][LMAX + 3][2 * LMAX + 5];
    pp->p = a + b;
    pp->cc = cc;
    for (int d = 0; d < 3; ++d) {
        pp->P[d] = (a * RA[d] + b * RB[d]) / pp->p;
        hermite_E(la, lb, a, b, RA[d] - RB[d], E);
        for (int i = 0; i <= la; ++i)
            for (int j = 0; j <= lb; ++j)
                for (int t = 0; t <= i + j; ++t) pp->E[d][i][j][t] = E[i][j][t];
    }
}

/* rotate one index of a 4-index Cartesian block to spherical: in (n0, nc, n2) -> out (n0, ns, n2) */
static void rot_axis(int l, int n0, int n2, const double *in, double *out)
{
    if (l <= 1) { /* s and p: the transformation is a scale (sph_matrix: T = c * identity), most shells are these */
        const double c = l == 0 ? 0.282094791773878143 : 0.488602511902919921;
        const size_t n = (size_t)n0 * (l == 0 ? 1 : 3) * n2;
        for (size_t i = 0; i < n; ++i) out[i] = c * in[i];
        return;
    }
    double T[7][MAXCART];
    sph_matrix(l, T);
    const int nc = NCART(l), ns = 2 * l + 1;
    for (int i0 = 0; i0 < n0; ++i0)
        for (int m = 0; m < ns; ++m)
            for (int i2 = 0; i2 < n2; ++i2) {
                double s = 0.0;
                for (int c = 0; c < nc; ++c)
                    if (T[m][c] != 0.0) s += T[m][c] * in[((size_t)i0 * nc + c) * n2 + i2];
                out[((size_t)i0 * ns + m) * n2 + i2] = s;
            }
}

/* Every shell pair A >= B with its primitive-pair data; shared by the dense and the column-wise
 * (Cholesky) drivers. */
typedef struct {
    int nshell, nao, npairs;
    const double *xyz, *ex, *cf;
    const int *ls, *nprim, *off, *ao0;
    int *pA, *pB;
    PrimPair **pairs;
    int *npp;     /* significant primitive pairs kept per shell pair */
    double *qmax; /* Schwarz bound sqrt(max (ab|ab)) per shell pair, filled by qc_eri_diag */
    /* owned copies of the shell table (the context outlives the caller's arrays) */
    double *own_xyz, *own_ex, *own_cf;
    int *own_i;
} EriCtx;

static int pair_index(int A, int B) { return A >= B ? A * (A + 1) / 2 + B : B * (B + 1) / 2 + A; }

static void ctx_build_pairs(EriCtx *c)
{
    const int nshell = c->nshell;
    c->npairs = nshell * (nshell + 1) / 2;
    c->pairs = (PrimPair **)calloc(c->npairs, sizeof(PrimPair *));
    c->npp = (int *)calloc(c->npairs, sizeof(int));
    c->pA = (int *)malloc(sizeof(int) * c->npairs);
    c->pB = (int *)malloc(sizeof(int) * c->npairs);
    c->qmax = NULL;
    for (int A = 0, k = 0; A < nshell; ++A)
        for (int B = 0; B <= A; ++B, ++k) { c->pA[k] = A; c->pB[k] = B; }
#pragma omp parallel for schedule(dynamic)
    for (int k = 0; k < c->npairs; ++k) {
        const int A = c->pA[k], B = c->pB[k];
        c->pairs[k] = (PrimPair *)malloc(sizeof(PrimPair) * c->nprim[A] * c->nprim[B]);
        /* primitive pairs whose Gaussian-product prefactor |c_a c_b| exp(-mu R_AB^2) is below 1e-18 are
         * dropped here once: tight primitives on different centres make up most of a contracted pair's
         * nprim[A]*nprim[B] products and contribute nothing at fp64 */
        const double *RA = c->xyz + 3 * A, *RB = c->xyz + 3 * B;
        const double R2 = (RA[0] - RB[0]) * (RA[0] - RB[0]) + (RA[1] - RB[1]) * (RA[1] - RB[1]) + (RA[2] - RB[2]) * (RA[2] - RB[2]);
        int kept = 0;
        for (int a = 0; a < c->nprim[A]; ++a)
            for (int b = 0; b < c->nprim[B]; ++b) {
                const double ea = c->ex[c->off[A] + a], eb = c->ex[c->off[B] + b];
                const double cc = c->cf[c->off[A] + a] * c->cf[c->off[B] + b];
                if (fabs(cc) * exp(-ea * eb / (ea + eb) * R2) < 1e-18) continue;
                make_pair(c->ls[A], c->ls[B], ea, eb, RA, RB, cc, &c->pairs[k][kept++]);
            }
        c->npp[k] = kept;
    }
}

static void ctx_free_pairs(EriCtx *c)
{
    for (int k = 0; k < c->npairs; ++k) free(c->pairs[k]);
    free(c->pairs); free(c->pA); free(c->pB); free(c->qmax); free(c->npp);
}

#define QUARTET_DOUBLES (MAXCART * MAXCART * MAXCART * MAXCART)

/* (AB|CD) for shell pairs kab = (A >= B), kcd = (C >= D): spherical block [nsa][nsb][nsc][nsd] left
 * in `cart` (t1 is scratch of the same size). */
static void quartet(const EriCtx *c, int kab, int kcd, double *cart, double *t1, double R[RDIM][RDIM][RDIM])
{
    const int A = c->pA[kab], B = c->pB[kab], C = c->pA[kcd], D = c->pB[kcd];
    const int *ls = c->ls;
    const int la = ls[A], lb = ls[B], lc = ls[C], ld = ls[D];
    const int nca = NCART(la), ncb = NCART(lb), ncc = NCART(lc), ncd = NCART(ld);
    int ax[MAXCART], ay[MAXCART], az[MAXCART], bx[MAXCART], by[MAXCART], bz[MAXCART];
    int cx[MAXCART], cy[MAXCART], cz[MAXCART], dx[MAXCART], dy[MAXCART], dz[MAXCART];
    cart_components(la, ax, ay, az); cart_components(lb, bx, by, bz);
    cart_components(lc, cx, cy, cz); cart_components(ld, dx, dy, dz);
    const int Lab = la + lb, Lcd = lc + ld, L = Lab + Lcd;
    (void)Lcd;
    memset(cart, 0, sizeof(double) * nca * ncb * ncc * ncd);
    const int nab = c->npp[kab], ncdp = c->npp[kcd];
    const double two_pi_52 = 34.986836655249725; /* 2 pi^(5/2) */
    /* Sparse Hermite lists.  A component pair's expansion E^x_t E^y_u E^z_v has at most (2+1)^3 = 27 terms
     * (l <= 3 per shell); forming the products once per primitive pair -- the ket side outside the loop over the bra
     * primitives, the bra side once per primitive quartet -- leaves the two contractions below as plain sums of
     * products (the straightforward form multiplied the three factors again for every (t,u,v) and every component
     * of the other side: 77 % of the engine's time under gprof; 1.45x faster on one thread this way). */
    enum { MAXE = 27 };
    static _Thread_local double wk[MAXCART * MAXCART][MAXE], wb[MAXCART * MAXCART][MAXE];
    static _Thread_local int ok_[MAXCART * MAXCART][MAXE], ob[MAXCART * MAXCART][MAXE], nk[MAXCART * MAXCART], nb[MAXCART * MAXCART];
    static _Thread_local double g[RDIM * RDIM * RDIM];
    const double *Rf = &R[0][0][0];
    /* bra-side (t,u,v) set: every t+u+v <= Lab, as flat offsets into R / g */
    int tuv[HDIM * HDIM * HDIM], ntuv = 0;
    for (int t = 0; t <= Lab; ++t)
        for (int u = 0; u <= Lab - t; ++u)
            for (int v = 0; v <= Lab - t - u; ++v) tuv[ntuv++] = (t * RDIM + u) * RDIM + v;
    for (int icd = 0; icd < ncdp; ++icd) {
        const PrimPair *cd = &c->pairs[kcd][icd];
        for (int ic = 0; ic < ncc; ++ic)
            for (int id = 0; id < ncd; ++id) {
                const int k = ic * ncd + id;
                int n = 0;
                for (int a1 = 0; a1 <= cx[ic] + dx[id]; ++a1) {
                    const double e1 = cd->E[0][cx[ic]][dx[id]][a1];
                    for (int a2 = 0; a2 <= cy[ic] + dy[id]; ++a2) {
                        const double e2 = e1 * cd->E[1][cy[ic]][dy[id]][a2];
                        for (int a3 = 0; a3 <= cz[ic] + dz[id]; ++a3) {
                            const double e3 = e2 * cd->E[2][cz[ic]][dz[id]][a3];
                            wk[k][n] = ((a1 + a2 + a3) & 1) ? -e3 : e3;
                            ok_[k][n] = (a1 * RDIM + a2) * RDIM + a3;
                            ++n;
                        }
                    }
                }
                nk[k] = n;
            }
        for (int iab = 0; iab < nab; ++iab) {
            const PrimPair *ab = &c->pairs[kab][iab];
            const double p = ab->p, q = cd->p, alpha = p * q / (p + q);
            const double PQ[3] = {ab->P[0] - cd->P[0], ab->P[1] - cd->P[1], ab->P[2] - cd->P[2]};
            hermite_R(L, alpha, PQ, R);
            const double pref = two_pi_52 / (p * q * sqrt(p + q)) * ab->cc * cd->cc;
            for (int ia = 0; ia < nca; ++ia)
                for (int ib = 0; ib < ncb; ++ib) {
                    const int k = ia * ncb + ib;
                    int n = 0;
                    for (int t = 0; t <= ax[ia] + bx[ib]; ++t) {
                        const double e1 = pref * ab->E[0][ax[ia]][bx[ib]][t];
                        for (int u = 0; u <= ay[ia] + by[ib]; ++u) {
                            const double e2 = e1 * ab->E[1][ay[ia]][by[ib]][u];
                            for (int v = 0; v <= az[ia] + bz[ib]; ++v) {
                                wb[k][n] = e2 * ab->E[2][az[ia]][bz[ib]][v];
                                ob[k][n] = (t * RDIM + u) * RDIM + v;
                                ++n;
                            }
                        }
                    }
                    nb[k] = n;
                }
            for (int kc = 0; kc < ncc * ncd; ++kc) {
                /* g[t][u][v] = sum_{tau,nu,phi} (-1)^(tau+nu+phi) E^cd R[t+tau][u+nu][v+phi] for this ket component */
                const int ne = nk[kc];
                for (int m = 0; m < ntuv; ++m) {
                    const double *Rm = Rf + tuv[m];
                    double sacc = 0.0;
                    for (int e = 0; e < ne; ++e) sacc += wk[kc][e] * Rm[ok_[kc][e]];
                    g[tuv[m]] = sacc;
                }
                const int ic = kc / ncd, id = kc - ic * ncd;
                for (int kb = 0; kb < nca * ncb; ++kb) {
                    double sacc = 0.0;
                    for (int e = 0; e < nb[kb]; ++e) sacc += wb[kb][e] * g[ob[kb][e]];
                    cart[((size_t)kb * ncc + ic) * ncd + id] += sacc;
                }
            }
        }
    }
    /* Cartesian -> spherical on the four indices */
    const int nsa = 2 * la + 1, nsb = 2 * lb + 1, nsc = 2 * lc + 1;
    rot_axis(la, 1, ncb * ncc * ncd, cart, t1);
    rot_axis(lb, nsa, ncc * ncd, t1, cart);
    rot_axis(lc, nsa * nsb, ncd, cart, t1);
    rot_axis(ld, nsa * nsb * nsc, 1, t1, cart);
}

int qc_int2e(int nshell, const double *xyz, const int *ls, const int *nprim, const int *off,
             const int *ao0, const double *ex, const double *cf, int nao, double *eri)
{
    for (int s = 0; s < nshell; ++s)
        if (ls[s] < 0 || ls[s] > LMAX) return -1;
    const size_t n = (size_t)nao;
    EriCtx ctx = {nshell, nao, 0, xyz, ex, cf, ls, nprim, off, ao0};
    ctx_build_pairs(&ctx);
#pragma omp parallel
    {
        double *cart = (double *)malloc(sizeof(double) * QUARTET_DOUBLES);
        double *t1 = (double *)malloc(sizeof(double) * QUARTET_DOUBLES);
        static _Thread_local double R[RDIM][RDIM][RDIM];
#pragma omp for schedule(dynamic)
        for (int kab = 0; kab < ctx.npairs; ++kab)
            for (int kcd = 0; kcd <= kab; ++kcd) {
                const int A = ctx.pA[kab], B = ctx.pB[kab], C = ctx.pA[kcd], D = ctx.pB[kcd];
                const int nsa = 2 * ls[A] + 1, nsb = 2 * ls[B] + 1, nsc = 2 * ls[C] + 1, nsd = 2 * ls[D] + 1;
                quartet(&ctx, kab, kcd, cart, t1, R);
                /* scatter with the 8-fold permutational symmetry */
                for (int a = 0; a < nsa; ++a)
                    for (int b = 0; b < nsb; ++b)
                        for (int c = 0; c < nsc; ++c)
                            for (int d = 0; d < nsd; ++d) {
                                const double v = cart[(((size_t)a * nsb + b) * nsc + c) * nsd + d];
                                const size_t i = ao0[A] + a, j = ao0[B] + b, k = ao0[C] + c, l = ao0[D] + d;
                                eri[((i * n + j) * n + k) * n + l] = v;
                                eri[((j * n + i) * n + k) * n + l] = v;
                                eri[((i * n + j) * n + l) * n + k] = v;
                                eri[((j * n + i) * n + l) * n + k] = v;
                                eri[((k * n + l) * n + i) * n + j] = v;
                                eri[((l * n + k) * n + i) * n + j] = v;
                                eri[((k * n + l) * n + j) * n + i] = v;
                                eri[((l * n + k) * n + j) * n + i] = v;
                            }
            }
        free(cart);
        free(t1);
    }
    ctx_free_pairs(&ctx);
    return 0;
}

/* ---- column-wise access for the pivoted Cholesky factorisation (no dense ERI) ------------------- */
void *qc_eri_open(int nshell, const double *xyz, const int *ls, const int *nprim, const int *off,
                  const int *ao0, const double *ex, const double *cf, int nao, int nprim_total)
{
    for (int s = 0; s < nshell; ++s)
        if (ls[s] < 0 || ls[s] > LMAX) return NULL;
    EriCtx *c = (EriCtx *)calloc(1, sizeof(EriCtx));
    c->nshell = nshell; c->nao = nao;
    c->own_xyz = (double *)malloc(sizeof(double) * 3 * nshell);
    c->own_ex = (double *)malloc(sizeof(double) * nprim_total);
    c->own_cf = (double *)malloc(sizeof(double) * nprim_total);
    c->own_i = (int *)malloc(sizeof(int) * 4 * nshell);
    memcpy(c->own_xyz, xyz, sizeof(double) * 3 * nshell);
    memcpy(c->own_ex, ex, sizeof(double) * nprim_total);
    memcpy(c->own_cf, cf, sizeof(double) * nprim_total);
    memcpy(c->own_i, ls, sizeof(int) * nshell);
    memcpy(c->own_i + nshell, nprim, sizeof(int) * nshell);
    memcpy(c->own_i + 2 * nshell, off, sizeof(int) * nshell);
    memcpy(c->own_i + 3 * nshell, ao0, sizeof(int) * nshell);
    c->xyz = c->own_xyz; c->ex = c->own_ex; c->cf = c->own_cf;
    c->ls = c->own_i; c->nprim = c->own_i + nshell; c->off = c->own_i + 2 * nshell; c->ao0 = c->own_i + 3 * nshell;
    ctx_build_pairs(c);
    return c;
}

void qc_eri_close(void *h)
{
    EriCtx *c = (EriCtx *)h;
    if (!c) return;
    ctx_free_pairs(c);
    free(c->own_xyz); free(c->own_ex); free(c->own_cf); free(c->own_i);
    free(c);
}

/* diag[i*nao + j] = (ij|ij); also records the Schwarz bounds used to skip quartets in qc_eri_cols */
int qc_eri_diag(void *h, double *diag)
{
    EriCtx *c = (EriCtx *)h;
    if (!c) return -1;
    const size_t n = (size_t)c->nao;
    if (!c->qmax) c->qmax = (double *)malloc(sizeof(double) * c->npairs);
#pragma omp parallel
    {
        double *cart = (double *)malloc(sizeof(double) * QUARTET_DOUBLES);
        double *t1 = (double *)malloc(sizeof(double) * QUARTET_DOUBLES);
        static _Thread_local double R[RDIM][RDIM][RDIM];
#pragma omp for schedule(dynamic)
        for (int kab = 0; kab < c->npairs; ++kab) {
            const int A = c->pA[kab], B = c->pB[kab];
            const int nsa = 2 * c->ls[A] + 1, nsb = 2 * c->ls[B] + 1;
            quartet(c, kab, kab, cart, t1, R);
            double m = 0.0;
            for (int a = 0; a < nsa; ++a)
                for (int b = 0; b < nsb; ++b) {
                    const double v = cart[(((size_t)a * nsb + b) * nsa + a) * nsb + b];
                    const size_t i = c->ao0[A] + a, j = c->ao0[B] + b;
                    diag[i * n + j] = v;
                    diag[j * n + i] = v;
                    if (v > m) m = v;
                }
            c->qmax[kab] = sqrt(m);
        }
        free(cart);
        free(t1);
    }
    return 0;
}

/* All integrals (ij|kl) with k in shell C, l in shell D: out[(k_local*nsd + l_local)][i][j], each an
 * (nao,nao) symmetric matrix.  Quartets below the Schwarz bound `screen` are left zero.
 * qc_eri_diag must have been called. */
int qc_eri_cols2(void *h, int C, int D, double screen, double *out, int lower_only);
int qc_eri_cols(void *h, int C, int D, double screen, double *out) { return qc_eri_cols2(h, C, D, screen, out, 0); }

/* lower_only: each matrix (..|kl) is written for i >= j only (the shell pairs are stored A >= B); the transposed
 * element is the caller's to fill -- the single-element strided stores of the mirror image cost as much as the
 * integrals themselves at nao ~ 500 (cache and TLB misses over up to 95 MB), and a GPU consumer mirrors for free. */
int qc_eri_cols2(void *h, int C, int D, double screen, double *out, int lower_only)
{
    EriCtx *c = (EriCtx *)h;
    if (!c || !c->qmax || C < 0 || D < 0 || C >= c->nshell || D >= c->nshell) return -1;
    const size_t n = (size_t)c->nao;
    const int kcd = pair_index(C, D);
    const int swap = C < D; /* stored pair is (max, min) */
    const int nsc = 2 * c->ls[C] + 1, nsd = 2 * c->ls[D] + 1;
    const size_t nout = n * n * nsc * nsd;
#pragma omp parallel
    {
        /* zero-fill in parallel: up to 49 matrices of nao^2 doubles (95 MB at nao 494), which one thread clears in
         * about the time sixteen take for the block's integrals */
#pragma omp for schedule(static)
        for (long blk = 0; blk < (long)((nout + 65535) / 65536); ++blk) {
            const size_t lo = (size_t)blk * 65536, hi = lo + 65536 < nout ? lo + 65536 : nout;
            memset(out + lo, 0, sizeof(double) * (hi - lo));
        }
        double *cart = (double *)malloc(sizeof(double) * QUARTET_DOUBLES);
        double *t1 = (double *)malloc(sizeof(double) * QUARTET_DOUBLES);
        static _Thread_local double R[RDIM][RDIM][RDIM];
#pragma omp for schedule(dynamic)
        for (int kab = 0; kab < c->npairs; ++kab) {
            if (c->qmax[kab] * c->qmax[kcd] < screen) continue;
            const int A = c->pA[kab], B = c->pB[kab];
            const int nsa = 2 * c->ls[A] + 1, nsb = 2 * c->ls[B] + 1;
            quartet(c, kab, kcd, cart, t1, R);
            /* block is [a][b][c'][d'] with (c',d') over the stored (max,min) shells */
            const int n3 = swap ? nsd : nsc, n4 = swap ? nsc : nsd;
            for (int a = 0; a < nsa; ++a)
                for (int b = 0; b < nsb; ++b)
                    for (int k = 0; k < nsc; ++k)
                        for (int l = 0; l < nsd; ++l) {
                            const int c3 = swap ? l : k, c4 = swap ? k : l;
                            const double v = cart[(((size_t)a * nsb + b) * n3 + c3) * n4 + c4];
                            const size_t i = c->ao0[A] + a, j = c->ao0[B] + b;
                            double *m = out + (size_t)(k * nsd + l) * n * n;
                            m[i * n + j] = v;
                            if (!lower_only) m[j * n + i] = v;
                        }
        }
        free(cart);
        free(t1);
    }
    return 0;
}

/* worker threads of the OpenMP regions above (default: every visible core, which oversubscribes a
 * cgroup CPU share) */
void qc_set_threads(int n)
{
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}
