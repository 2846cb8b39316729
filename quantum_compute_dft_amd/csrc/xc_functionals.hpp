// Pointwise exchange-correlation functionals, fp64, closed shell.
//
// Device restatement of the ten __device__ functionals of the reference
// (src/dft_solver.cu:61-283) and of the per-point bodies of its three fused
// kernels (:309-344, :382-432, :434-513).  Same constants, same density /
// gradient cut-offs, same algebra; `quirks` selects the reference's shipped
// derivative formulas (true) or the finite-difference-verified ones (false)
// for the two places they differ.
#pragma once
#include <hip/hip_runtime.h>

namespace qcdft {
namespace xc {

constexpr double kRhoCut = 1e-12;   // src/dft_solver.cu:12
constexpr double kSigmaCut = 1e-20; // src/dft_solver.cu:13
constexpr double kPi = 3.14159265358979323846;
constexpr double kCx = 0.7385587663820224;

struct Lda { double e, v; };
struct Gga { double e, vr, vs; };

// Slater exchange, src/dft_solver.cu:61-76 (both spellings give the same values).
__device__ __forceinline__ Lda slater_x(double rho)
{
    if (rho < kRhoCut) return {0.0, 0.0};
    double e = -kCx * cbrt(rho);
    return {e, (4.0 / 3.0) * e};
}

// Shared VWN form: eps(x) and d eps/dx for parameters (A,b,c,x0).
// `with_atan_terms` = false reproduces src/dft_solver.cu:192-193.
__device__ __forceinline__ void vwn_form(double x, double A, double b, double c, double x0,
                                         bool with_atan_terms, double &eps, double &deps_dx)
{
    const double X = x * x + b * x + c;
    const double Q = sqrt(4.0 * c - b * b);
    const double X0 = x0 * x0 + b * x0 + c;
    const double at = atan(Q / (2.0 * x + b));
    const double lg = log(x * x / X);
    const double lg0 = log((x - x0) * (x - x0) / X);
    const double w0 = b * x0 / X0;
    eps = A * (lg + (2.0 * b / Q) * at - w0 * (lg0 + (2.0 * (2.0 * x0 + b) / Q) * at));
    const double dl = 2.0 / x - (2.0 * x + b) / X;
    const double dl0 = 2.0 / (x - x0) - (2.0 * x + b) / X;
    if (with_atan_terms)
        deps_dx = A * (dl - b / X - w0 * (dl0 - (2.0 * x0 + b) / X));
    else
        deps_dx = A * (dl - w0 * dl0);
}

// VWN5 paramagnetic, src/dft_solver.cu:180-205 (parameters :21-24).
__device__ __forceinline__ Lda vwn5_c(double rho, bool quirks)
{
    if (rho < kRhoCut) return {0.0, 0.0};
    const double rs = cbrt(3.0 / (4.0 * kPi * rho));
    const double x = sqrt(rs);
    double e, de;
    vwn_form(x, 0.0310907, 3.72744, 12.9352, -0.10498, !quirks, e, de);
    return {e, e - (rs / 3.0) * (de / (2.0 * x))};
}

// VWN-RPA as used by B3LYP, src/dft_solver.cu:106-138 (parameters :38-41).
__device__ __forceinline__ Lda vwn_rpa_c(double rho)
{
    if (rho < kRhoCut) return {0.0, 0.0};
    const double rs = cbrt(3.0 / (4.0 * kPi * rho));
    const double x = sqrt(rs);
    double e, de;
    vwn_form(x, 0.0310907, 13.0720, 42.7198, -0.409286, true, e, de);
    return {e, e - (rs / 3.0) * (de / (2.0 * x))};
}

// PW92 (modified), src/dft_solver.cu:207-220 (parameters :25-31).
__device__ __forceinline__ Lda pw92_c(double rho)
{
    if (rho < kRhoCut) return {0.0, 0.0};
    constexpr double A = 0.03109069086965489503;
    constexpr double a1 = 0.21370, b1 = 7.5957, b2 = 3.5876, b3 = 1.6382, b4 = 0.49294;
    const double rs = cbrt(3.0 / (4.0 * kPi * rho));
    const double sq = sqrt(rs);
    const double Q = 2.0 * A * (b1 * sq + b2 * rs + b3 * rs * sq + b4 * rs * rs);
    const double Qp = 2.0 * A * (0.5 * b1 / sq + b2 + 1.5 * b3 * sq + 2.0 * b4 * rs);
    const double lg = log(1.0 + 1.0 / Q);
    const double f = -2.0 * A * (1.0 + a1 * rs);
    const double e = f * lg;
    const double de = -2.0 * A * a1 * lg + f * (1.0 / (1.0 + 1.0 / Q)) * (-1.0 / (Q * Q)) * Qp;
    return {e, e - (rs / 3.0) * de};
}

// PBE exchange, src/dft_solver.cu:222-242.
__device__ __forceinline__ Gga pbe_x(double rho, double sigma)
{
    if (rho < kRhoCut) return {0.0, 0.0, 0.0};
    constexpr double kappa = 0.804, mu = 0.2195149727645171;
    const double r13 = cbrt(rho);
    const double r43 = rho * r13;
    const double kF = cbrt(3.0 * kPi * kPi * rho);
    const double den = 4.0 * kF * kF * rho * rho;
    double s2 = 0.0;
    if (sigma > kSigmaCut && den > 1e-50) s2 = sigma / den;
    if (s2 > 1e12) s2 = 1e12;
    const double num = 1.0 + mu * s2 / kappa;
    const double F = 1.0 + kappa * (1.0 - 1.0 / num);
    const double e = -kCx * r13 * F;
    const double dF = mu / (num * num);
    Gga o;
    o.e = e;
    o.vs = (-kCx * r43) * dF * (1.0 / den);
    o.vr = (4.0 / 3.0) * e - (8.0 / 3.0) * (-kCx * r43) * s2 * dF / rho;
    return o;
}

// PBE correlation, src/dft_solver.cu:244-283.
__device__ __forceinline__ Gga pbe_c(double rho, double sigma, bool quirks)
{
    if (rho < kRhoCut) return {0.0, 0.0, 0.0};
    constexpr double beta = 0.066725, gamma = 0.03109069086965489503;
    const Lda l = pw92_c(rho);
    const double kF = cbrt(3.0 * kPi * kPi * rho);
    const double den16 = 16.0 * kF * rho * rho;
    double t2 = 0.0;
    if (sigma > kSigmaCut && den16 > 1e-50) t2 = (sigma * kPi) / den16;
    if (t2 > 1.0e20) t2 = 1.0e20;
    const double x = -l.e / gamma;
    const double em1 = expm1(x);
    const double A = (fabs(em1) < 1e-20) ? 1.0e20 : (beta / gamma) / em1;
    const double At2 = A * t2;
    const double num = 1.0 + At2;
    const double den = 1.0 + At2 + At2 * At2;
    const double Qr = num / den;
    const double tl = 1.0 + (beta / gamma) * t2 * Qr;
    const double H = gamma * log(tl);
    const double Qp = (den - num * (1.0 + 2.0 * At2)) / (den * den);
    const double pre = gamma / tl * (beta / gamma);
    const double dH_dt2 = pre * (Qr + At2 * Qp);
    const double dH_dA = pre * t2 * t2 * Qp;
    const double dt2_ds = (den16 > 1e-50) ? kPi / den16 : 0.0;
    double dx_drho = (l.v - l.e) / (rho * gamma); // :277 as shipped
    if (!quirks) dx_drho = -dx_drho;              // x = -ec/gamma
    const double dA_drho = (-A * exp(x) / em1) * dx_drho;
    const double dt2_drho = t2 * (-7.0 / 3.0) / rho;
    Gga o;
    o.e = l.e + H;
    o.vs = rho * dH_dt2 * dt2_ds;
    o.vr = l.v + H + rho * (dH_dA * dA_drho + dH_dt2 * dt2_drho);
    return o;
}

// Becke-88 gradient correction, per-spin arguments, src/dft_solver.cu:78-104.
__device__ __forceinline__ Gga b88_x(double rho, double sigma)
{
    if (rho < kRhoCut || sigma < kSigmaCut) return {0.0, 0.0, 0.0};
    constexpr double beta = 0.0042; // :43
    const double r13 = cbrt(rho);
    const double r43 = rho * r13;
    const double g = sqrt(sigma);
    const double x = g / r43;
    const double x2 = x * x;
    const double as = asinh(x);
    const double den = 1.0 + 6.0 * beta * x * as;
    const double term = beta * x2 / den;
    const double dden = 6.0 * beta * (as + x / sqrt(1.0 + x2));
    const double dF = beta * (2.0 * x * den - x2 * dden) / (den * den);
    const double dE = r43 * (-dF);
    Gga o;
    o.e = -term * r13;
    o.vs = dE * (1.0 / (2.0 * r43 * g));
    o.vr = (4.0 / 3.0) * ((r43 * (-term)) / rho) - (4.0 / 3.0) * dE * (x / rho);
    return o;
}

// Closed-shell LYP, src/dft_solver.cu:140-178 (constants :45-49).
__device__ __forceinline__ Gga lyp_c(double rho, double sigma)
{
    if (rho < 1e-14) return {0.0, 0.0, 0.0};
    constexpr double a = 0.04918, b = 0.132, c = 0.2533, d = 0.349;
    constexpr double CF = 2.87123400018819108;
    const double rm13 = 1.0 / cbrt(rho);
    const double rm53 = rm13 * rm13 * rm13 * rm13 * rm13;
    const double ev = exp(-c * rm13);
    const double den = 1.0 + d * rm13;
    const double di = 1.0 / den;
    const double G = ev * di;
    const double delta = c * rm13 + d * rm13 * di;
    const double gb = 3.0 + 7.0 * delta;
    const double k72 = a * b / 72.0;
    const double H = -a * rho * di - a * b * CF * rho * G + k72 * sigma * rm53 * G * gb;
    const double d_rm13 = -(1.0 / 3.0) * rm13 / rho;
    const double d_den = d * d_rm13;
    const double d_G = G * delta / (3.0 * rho);
    const double d_delta = c * d_rm13 + d * (d_rm13 * di - rm13 * di * di * d_den);
    const double d_H1 = -a * (den - rho * d_den) * (di * di);
    const double d_H2a = -a * b * CF * (G + rho * d_G);
    const double tdv = (-5.0 / (3.0 * rho)) * gb + (delta / (3.0 * rho)) * gb + 7.0 * d_delta;
    Gga o;
    o.e = H / rho;
    o.vr = d_H1 + d_H2a + k72 * sigma * (rm53 * G) * tdv;
    o.vs = k72 * rm53 * G * gb;
    return o;
}

// What one grid point contributes: the energy density rho*eps and the four
// coefficients of B[g,:] = c0*phi + c1*dphi/dx + c2*dphi/dy + c3*dphi/dz.
struct PointXC { double exc, c0, c1, c2, c3; };

// lda_fused_kernel body, src/dft_solver.cu:317-342.
__device__ __forceinline__ PointXC lda_point(double rho, double w, bool quirks)
{
    if (rho < kRhoCut) return {0.0, 0.0, 0.0, 0.0, 0.0};
    const Lda x = slater_x(rho), c = vwn5_c(rho, quirks);
    return {rho * (x.e + c.e), w * (x.v + c.v), 0.0, 0.0, 0.0};
}

// gga_fused_kernel body, src/dft_solver.cu:391-430 (factor 4 at :429).
__device__ __forceinline__ PointXC gga_point(double rho, double sigma, double gx, double gy,
                                             double gz, double w, bool quirks)
{
    if (rho < kRhoCut) return {0.0, 0.0, 0.0, 0.0, 0.0};
    const Gga x = pbe_x(rho, sigma), c = pbe_c(rho, sigma, quirks);
    const double f = w * 4.0 * (x.vs + c.vs);
    return {rho * (x.e + c.e), w * (x.vr + c.vr), f * gx, f * gy, f * gz};
}

// b3lyp_fused_kernel body, src/dft_solver.cu:444-511 (mixing :33-36, the 0.5
// of :468 and :492, factor 2 at :510).
__device__ __forceinline__ PointXC b3lyp_point(double rho, double sigma, double gx, double gy,
                                               double gz, double w)
{
    if (rho < kRhoCut) return {0.0, 0.0, 0.0, 0.0, 0.0};
    constexpr double cL = 0.80, cB = 0.72, cV = 0.19, cY = 0.81;
    const Lda xl = slater_x(rho);
    Gga xb = b88_x(0.5 * rho, 0.25 * sigma);
    xb.vs *= 0.5;
    const Lda cv = vwn_rpa_c(rho);
    const Gga cy = lyp_c(rho, sigma);
    const double eps = cL * xl.e + cB * xb.e + cV * cv.e + cY * cy.e;
    const double vr = 0.5 * (cL * xl.v + cB * xb.vr + cV * cv.v + cY * cy.vr);
    const double f = w * 2.0 * (cB * xb.vs + cY * cy.vs);
    return {rho * eps, w * vr, f * gx, f * gy, f * gz};
}

} // namespace xc
} // namespace qcdft
