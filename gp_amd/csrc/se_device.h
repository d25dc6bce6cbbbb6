// Device functions of the covariance builders shared by se_kernels.hip (one tile per workgroup of a big
// launch) and chol_kernels.hip (the one-workgroup small-N evaluation builds its matrix with the SAME
// arithmetic, so both paths produce bit-identical covariance matrices).  gfx950 only.
// Reference formulas: R/kernels.R:11-24 (QQ, QQard), Stan cov_exp_quad (models/fit_hyperparameters.stan:19)
// and the diagonal update :21-24.
#pragma once
#include "gpmi_internal.h"

constexpr int SE_TR = 64, SE_TC = 8 * (512 / SE_TR);  // tile of se_cov_tile: 256 threads x (2 rows, 8 columns)

template <bool NT = false>
__device__ __forceinline__ void store_pair(double *p, double v0, double v1, bool ok0, bool ok1, bool vec)
{
    if (ok0 && ok1 && vec) {
        if constexpr (NT) {  // streaming store: +8 % for the SE build, -10 % for the joint build (measured)
            typedef double dv2 __attribute__((ext_vector_type(2)));
            dv2 v = {v0, v1};
            __builtin_nontemporal_store(v, reinterpret_cast<dv2 *>(p));
        } else {
            *reinterpret_cast<double2 *>(p) = make_double2(v0, v1);
        }
    } else {
        if (ok0) p[0] = v0;
        if (ok1) p[1] = v1;
    }
}

// exp(x) for x <= 0 (the argument of the squared-exponential kernel).  Argument reduction
// x = n ln2 + r, |r| <= ln2 / 2, Taylor polynomial of degree 13 in r (remainder < 5e-18
// relative), result = ldexp(p, n); <= 1 ulp of error like the library routine.  What it drops is
// the library's overflow / infinity handling (x > 0 never happens; x < -800 is clamped and
// underflows to 0 through ldexp) and, above all, its code shape: the coefficients come from
// constant memory into SGPRs and every Horner step is ONE v_fma_f64 -- the library's fmac form
// needs two v_mov_b32 per coefficient and element (19 of 60 VALU instructions per element of
// the build kernel), which made an HBM-write-bound kernel half compute-bound.
struct ExpC {
    double c[18];
};
// passed as a kernel argument: values the compiler cannot see stay in SGPRs (constants it can see
// are re-materialised with v_mov in front of every fmac)
static const ExpC h_exp __attribute__((unused)) = {{
    1.4426950408889634074,        // log2(e)
    -6.93147180369123816490e-01,  // -ln2 high part (32 trailing zero bits: n * hi is exact)
    -1.90821492927058770002e-10,  // -ln2 low part
    1.0 / 6227020800.0, 1.0 / 479001600.0, 1.0 / 39916800.0, 1.0 / 3628800.0, 1.0 / 362880.0, 1.0 / 40320.0,
    1.0 / 5040.0, 1.0 / 720.0, 1.0 / 120.0, 1.0 / 24.0, 1.0 / 6.0, 0.5, 1.0, 1.0,
    -800.0}};
__device__ __forceinline__ double exp_nonpos(double xin, const ExpC &e)
{
    double x = fmax(xin, e.c[17]);  // (v_max_f64 drops a NaN operand: put back by the select at the end)
    const double n = rint(x * e.c[0]);
    double r = fma(n, e.c[1], x);
    r = fma(n, e.c[2], r);
    double p = e.c[3];
#pragma unroll
    for (int k = 4; k <= 16; ++k) p = fma(p, r, e.c[k]);
    // a NaN coordinate gives a NaN covariance, like exp(NaN) in R and the not_nan check of
    // Stan's cov_exp_quad -- the factorisation then reports "not positive definite" instead of
    // silently treating the point as infinitely far away
    // (only a real NaN: exp(-inf) stays 0 as in R / Stan -- a squared scaled distance that overflows, or one
    // infinite coordinate, is a point infinitely far away, not an error)
    const double v = ldexp(p, (int)n);
    return (xin != xin) ? xin : v;
}

// K[i,j] = a2 * exp(-1/2 sum_d ((X[i,d]-Y[j,d]) * inv_ell[d])^2), diag_add on i == j if same.
// One SE_TR x SE_TC tile at (row0, col0) by the 256 threads of a workgroup.  P: anything with .a2, .inv_ell[d], .D
// SCALED: X, Y hold the coordinates already multiplied by the inverse length-scales (__dmul_rn, the same rounding)
template <int DT, bool SCALED = false, class P>
__device__ __forceinline__ void se_cov_tile(const double *__restrict__ X, int n, int ldx,
                                            const double *__restrict__ Y, int m, int ldy,
                                            const P &p, double diag_add, int same, int lower,
                                            double *__restrict__ K, size_t ldk, int vec, const ExpC &ec, int row0, int col0)
{
    const int D = (DT > 0) ? DT : p.D;
    // SE_TR x SE_TC = 64 x 64 tile per workgroup.  Measured (N = 16384, lower triangle, 1.07 GB, with
    // the lean exp above): 64 x 64 0.220 ms = 4.89 TB/s; 128 x 32 0.237; 32 x 128 0.232; 512 x 8
    // (4 KiB contiguous per column) 0.310; a 1-D grid over the lower-triangular tiles only 0.237 -
    // 0.280 depending on the walk.  With the library exp the 64 x 64 tile took 0.228 ms; with
    // non-temporal 16-B stores (store_pair) it takes 0.204 ms = 5.25 TB/s (128 x 32: 0.227).
    if (lower && col0 > row0 + SE_TR - 1) return;  // tile strictly above the diagonal
    const int tx = threadIdx.x & (SE_TR / 2 - 1), ty = threadIdx.x / (SE_TR / 2);
    const int r = row0 + 2 * tx;
    const bool ok0 = r < n, ok1 = r + 1 < n;
    double x0[GPMI_MAXD], x1[GPMI_MAXD];
#pragma unroll
    for (int d = 0; d < GPMI_MAXD; ++d) {
        if (d < D) {
            // __dmul_rn: keep the scaling a separate rounding so that (x_i - x_j) and (x_j - x_i)
            // are exact negatives (no fma contraction) and K comes out bit-symmetric
            if (SCALED) {
                x0[d] = ok0 ? X[(size_t)r + (size_t)d * ldx] : 0.0;
                x1[d] = ok1 ? X[(size_t)r + 1 + (size_t)d * ldx] : 0.0;
            } else {
                x0[d] = ok0 ? __dmul_rn(X[(size_t)r + (size_t)d * ldx], p.inv_ell[d]) : 0.0;
                x1[d] = ok1 ? __dmul_rn(X[(size_t)r + 1 + (size_t)d * ldx], p.inv_ell[d]) : 0.0;
            }
        }
    }
    // Interior tiles (wholly inside the matrix and, for the lower triangle, wholly below the diagonal -- 98 % of the tiles
    // at N = 16384): no bounds, no diagonal, no per-element predicates around the stores.  The build is VALU-bound, not
    // HBM-bound (~80 instructions per element, 30 of them double-precision, against 8 bytes stored), and a quarter of the
    // instructions were exec-mask bookkeeping of the guarded form.
    if (vec && row0 + SE_TR <= n && col0 + SE_TC <= m && (!same || col0 + SE_TC <= row0)) {  // workgroup-uniform
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int c = col0 + ty * 8 + q;
            double s0 = 0.0, s1 = 0.0;
#pragma unroll
            for (int d = 0; d < GPMI_MAXD; ++d) {
                if (d < D) {
                    const double yv = SCALED ? Y[(size_t)c + (size_t)d * ldy] : __dmul_rn(Y[(size_t)c + (size_t)d * ldy], p.inv_ell[d]);
                    const double d0 = __dsub_rn(x0[d], yv), d1 = __dsub_rn(x1[d], yv);
                    s0 = fma(d0, d0, s0);
                    s1 = fma(d1, d1, s1);
                }
            }
            const double v0 = p.a2 * exp_nonpos(-0.5 * s0, ec), v1 = p.a2 * exp_nonpos(-0.5 * s1, ec);
            if (vec & 2) store_pair<true>(K + (size_t)r + (size_t)c * ldk, v0, v1, true, true, true);
            else store_pair<false>(K + (size_t)r + (size_t)c * ldk, v0, v1, true, true, true);
        }
        return;
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int c = col0 + ty * 8 + q;
        if (c >= m) break;
        double s0 = 0.0, s1 = 0.0;
#pragma unroll
        for (int d = 0; d < GPMI_MAXD; ++d) {
            if (d < D) {
                const double yv = SCALED ? Y[(size_t)c + (size_t)d * ldy] : __dmul_rn(Y[(size_t)c + (size_t)d * ldy], p.inv_ell[d]);
                const double d0 = __dsub_rn(x0[d], yv), d1 = __dsub_rn(x1[d], yv);
                s0 = fma(d0, d0, s0);
                s1 = fma(d1, d1, s1);
            }
        }
        double v0 = p.a2 * exp_nonpos(-0.5 * s0, ec), v1 = p.a2 * exp_nonpos(-0.5 * s1, ec);
        if (same) {
            if (r == c) v0 = p.a2 + diag_add;  // Stan: diagonal exactly alpha^2 (+ sigma^2)
            if (r + 1 == c) v1 = p.a2 + diag_add;
        }
        const bool w0 = ok0 && (!lower || r >= c), w1 = ok1 && (!lower || r + 1 >= c);
        if (vec & 2) store_pair<true>(K + (size_t)r + (size_t)c * ldk, v0, v1, w0, w1, true);
        else store_pair<false>(K + (size_t)r + (size_t)c * ldk, v0, v1, w0, w1, vec != 0);
    }
}

// derivative_kernels.R:39-73 with unit amplitude; r = tj - tk, e = exp(-r^2/(2 l^2)).
__device__ __forceinline__ double deriv_val(int kind, double tj, double tk, double l2)
{
    if (kind == GPMI_RQ || kind == GPMI_TQ || kind == GPMI_TR) {  // :47-49, :59-61, :67-69
        const double t = tj; tj = tk; tk = t;
        kind -= 1;
    }
    const double r = tj - tk;
    const double e = exp(-(r * r / (2 * l2)));
    switch (kind) {
    case GPMI_QQ: return e;                                                       // :39-41
    case GPMI_QR: return (e * r) / l2;                                            // :43-45
    case GPMI_RR: return e / l2 - (e * r * r) / (l2 * l2);                        // :51-53
    case GPMI_QT: return -(e / l2) + (e * r * r) / (l2 * l2);                     // :55-57
    case GPMI_RT: return (3 * e * r) / (l2 * l2) - (e * r * r * r) / (l2 * l2 * l2);  // :63-65
    default:      // GPMI_TT :71-73
        return (3 * e) / (l2 * l2) - (6 * e * r * r) / (l2 * l2 * l2) +
               (e * r * r * r * r) / (l2 * l2 * l2 * l2);
    }
}

// alpha^2 times a derivative kernel; compat (R/kernels.R:31 as written): for RR alpha^2 multiplies the first term only
__device__ __forceinline__ double deriv_cov_val(int kind, int compat, double a2, double x, double yv, double l2)
{
    if (compat && kind == GPMI_RR) {
        const double r = x - yv;
        const double e = exp(-(r * r / (2 * l2)));
        return a2 * e / l2 - (e * r * r) / (l2 * l2);
    }
    return a2 * deriv_val(kind, x, yv, l2);
}
