/*
 * oracle/ao_oracle.c -- TEST INFRASTRUCTURE ONLY.  NOT PART OF THE PRODUCT.
 *
 * CPU restatement of what the reference obtains from PySCF at grid.py:38
 * (dft.numint.eval_ao(mol, coords, deriv=0)) and grid.py:30-31 (deriv=1,
 * planes [1:4]): contracted real-spherical Gaussian AO values and their
 * Cartesian gradients on a set of points.
 *
 * PARITY UNPINNED: the arithmetic lives in third-party PySCF/libcint (version
 * not pinned by the reference, README.md:31-36; not installed here) and the
 * reference holds no test vector at this boundary.  This file restates the
 * published conventions: phi = R(r) * S_lm, R = sum_p c_p
 * exp(-a_p r^2) with normalised coefficients supplied by the caller, S_lm real
 * solid harmonics ordered p: x,y,z; d: xy,yz,z2,xz,x2-y2; f: m=-3..3.
 * It is checked offline by tests/test_ao_oracle.py (finite differences,
 * quadrature normalisation, orthogonality of the 2l+1 components).
 */
#include <math.h>
#include <stddef.h>

/* value and gradient of the 2l+1 real solid harmonics at (x,y,z) */
static int solid_harmonics(int l, double x, double y, double z, double *S, double *Sx,
                           double *Sy, double *Sz)
{
    if (l == 0) {
        S[0] = 0.282094791773878143; Sx[0] = Sy[0] = Sz[0] = 0.0;
        return 1;
    }
    if (l == 1) {
        const double c = 0.488602511902919921;
        S[0] = c * x; Sx[0] = c; Sy[0] = 0; Sz[0] = 0;
        S[1] = c * y; Sx[1] = 0; Sy[1] = c; Sz[1] = 0;
        S[2] = c * z; Sx[2] = 0; Sy[2] = 0; Sz[2] = c;
        return 3;
    }
    if (l == 2) {
        const double c = 1.092548430592079070, d = 0.315391565252520002, e = 0.546274215296039535;
        S[0] = c * x * y; Sx[0] = c * y; Sy[0] = c * x; Sz[0] = 0;
        S[1] = c * y * z; Sx[1] = 0; Sy[1] = c * z; Sz[1] = c * y;
        S[2] = d * (2 * z * z - x * x - y * y); Sx[2] = -2 * d * x; Sy[2] = -2 * d * y; Sz[2] = 4 * d * z;
        S[3] = c * x * z; Sx[3] = c * z; Sy[3] = 0; Sz[3] = c * x;
        S[4] = e * (x * x - y * y); Sx[4] = 2 * e * x; Sy[4] = -2 * e * y; Sz[4] = 0;
        return 5;
    }
    if (l == 3) {
        const double f3 = 0.590043589926643510, f2 = 2.890611442640554055,
                     f1 = 0.457045799464465739, f0 = 0.373176332590115391,
                     f2b = 1.445305721320277020;
        const double xx = x * x, yy = y * y, zz = z * z;
        S[0] = f3 * y * (3 * xx - yy); Sx[0] = f3 * 6 * x * y; Sy[0] = f3 * (3 * xx - 3 * yy); Sz[0] = 0;
        S[1] = f2 * x * y * z; Sx[1] = f2 * y * z; Sy[1] = f2 * x * z; Sz[1] = f2 * x * y;
        S[2] = f1 * y * (4 * zz - xx - yy); Sx[2] = -2 * f1 * x * y; Sy[2] = f1 * (4 * zz - xx - 3 * yy); Sz[2] = 8 * f1 * y * z;
        S[3] = f0 * z * (2 * zz - 3 * xx - 3 * yy); Sx[3] = -6 * f0 * x * z; Sy[3] = -6 * f0 * y * z; Sz[3] = f0 * (6 * zz - 3 * xx - 3 * yy);
        S[4] = f1 * x * (4 * zz - xx - yy); Sx[4] = f1 * (4 * zz - 3 * xx - yy); Sy[4] = -2 * f1 * x * y; Sz[4] = 8 * f1 * x * z;
        S[5] = f2b * z * (xx - yy); Sx[5] = 2 * f2b * x * z; Sy[5] = -2 * f2b * y * z; Sz[5] = f2b * (xx - yy);
        S[6] = f3 * x * (xx - 3 * yy); Sx[6] = f3 * (3 * xx - 3 * yy); Sy[6] = -6 * f3 * x * y; Sz[6] = 0;
        return 7;
    }
    return 0;
}

/* ao (ngrid,nao); grad (3,ngrid,nao) or NULL.  Shell table as DFT_EvalAO. */
int orc_eval_ao(long ngrid, int nao, int nshell, const double *shl_xyz, const int *shl_l,
                const int *shl_nprim, const int *shl_off, const int *shl_ao,
                const double *prim_exp, const double *prim_coef, const double *coords,
                double *ao, double *grad)
{
    const size_t plane = (size_t)ngrid * nao;
    for (long g = 0; g < ngrid; ++g) {
        for (int s = 0; s < nshell; ++s) {
            const double x = coords[3 * g] - shl_xyz[3 * s];
            const double y = coords[3 * g + 1] - shl_xyz[3 * s + 1];
            const double z = coords[3 * g + 2] - shl_xyz[3 * s + 2];
            const double r2 = x * x + y * y + z * z;
            double R0 = 0.0, R1 = 0.0;
            for (int p = 0; p < shl_nprim[s]; ++p) {
                const double a = prim_exp[shl_off[s] + p];
                const double e = prim_coef[shl_off[s] + p] * exp(-a * r2);
                R0 += e;
                R1 -= 2.0 * a * e;
            }
            double S[7], Sx[7], Sy[7], Sz[7];
            const int nf = solid_harmonics(shl_l[s], x, y, z, S, Sx, Sy, Sz);
            if (nf == 0) return -1;
            for (int m = 0; m < nf; ++m) {
                const size_t o = (size_t)g * nao + shl_ao[s] + m;
                ao[o] = R0 * S[m];
                if (grad) {
                    grad[o] = R0 * Sx[m] + R1 * S[m] * x;
                    grad[plane + o] = R0 * Sy[m] + R1 * S[m] * y;
                    grad[2 * plane + o] = R0 * Sz[m] + R1 * S[m] * z;
                }
            }
        }
    }
    return 0;
}
