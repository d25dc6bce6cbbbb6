// Covariance-matrix builders for gfx950: HBM-write-bound kernels.
//
// One workgroup (256 threads = 4 wave64) owns a 64-row x 64-column tile of the
// column-major output.  Lane tx (0..31) owns two consecutive rows, so a wave
// writes 2 columns x 512 contiguous bytes per store instruction (whole 128-B
// lines, 16 B per lane); the 8 column groups of the workgroup walk 8 columns
// each.  The input points are tiny (n x D doubles) and stay in L1/L2: the row
// coordinates live in registers, the column coordinates are wave-uniform
// broadcast loads.  One exp per element; for the joint [value; derivative]
// matrix one exp feeds all four blocks.
//
// Reference formulas: R/kernels.R:19-32, derivative_kernels.R:39-73,
// R/ode_gp_library.R:29-30, Stan cov_exp_quad (models/fit_hyperparameters.stan:19).
#include "gpmi_internal.h"

namespace {
#include "se_device.h"

constexpr int TILE = 64;

template <int DT>
__global__ __launch_bounds__(256) void k_se_cov(const double *__restrict__ X, int n, int ldx,
                                                const double *__restrict__ Y, int m, int ldy,
                                                SeParams p, double diag_add, int same, int lower,
                                                double *__restrict__ K, size_t ldk, int vec, ExpC ec)
{
    se_cov_tile<DT>(X, n, ldx, Y, m, ldy, p, diag_add, same, lower, K, ldk, vec, ec, (int)blockIdx.x * SE_TR, (int)blockIdx.y * SE_TC);
}

// Any D (R's QQard takes any D, R/kernels.R:11-19; D = 1, 2, 3 have the register-resident kernel
// above): LDS-tiled pairwise distances.  A workgroup owns 128 rows x 64 columns; the scaled
// coordinates of those rows and columns are staged in LDS in chunks of BIG_DC dimensions (each
// (operand, dimension) row of 64 values is loaded and scaled by ONE wave: the dimension index stays
// wave-uniform, so the inverse length-scales come out of the kernel arguments by scalar loads).  A
// thread keeps the 32 running sums of its 4 rows x 8 columns in registers (rows 2 tx, 2 tx + 1 and the
// same pair 64 rows further down: every 16-B store instruction of a wave then covers 512 contiguous
// bytes of a column; four consecutive rows per lane leave every 64-B segment half written and cost 2x)
// and reads, per dimension,
// its 4 row values (two 16-B LDS reads) and 8 column values (four 16-B broadcast reads): 2 VALU
// instructions and 0.19 LDS reads per element and dimension -- with 2 x 8 elements per thread the
// LDS pipe (5 reads per 16 elements) was the bound, not the VALU.  Compute-bound from D ~ 8 on
// (D = 64: 128 of ~170 VALU instructions per element are the distance); measured rates in DESIGN.md
// section 5.
constexpr int BIG_DC = 16, BIG_TR = 128, BIG_TC = 64;
__global__ __launch_bounds__(256) void k_se_cov_big(const double *__restrict__ X, int n, int ldx,
                                                    const double *__restrict__ Y, int m, int ldy,
                                                    SeParams p, double diag_add, int same, int lower,
                                                    double *__restrict__ K, size_t ldk, int vec, ExpC ec)
{
    __shared__ __attribute__((aligned(16))) double sX[BIG_DC][BIG_TR];
    __shared__ __attribute__((aligned(16))) double sY[BIG_DC][BIG_TC];
    const int row0 = blockIdx.x * BIG_TR, col0 = blockIdx.y * BIG_TC;
    if (lower && col0 > row0 + BIG_TR - 1) return;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = row0 + 2 * tx;   // this lane's rows: r, r + 1, r + 64, r + 65
    // staging sources of this lane, clamped (rows / columns past the end are never stored)
    const int xr0 = (row0 + lane < n) ? row0 + lane : n - 1;
    const int xr1 = (row0 + 64 + lane < n) ? row0 + 64 + lane : n - 1;
    const int yc = (col0 + lane < m) ? col0 + lane : m - 1;
    double acc[4][8];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int q = 0; q < 8; ++q) acc[a][q] = 0.0;
    for (int d0 = 0; d0 < p.D; d0 += BIG_DC) {
        const int dc = (p.D - d0 < BIG_DC) ? p.D - d0 : BIG_DC;
        __syncthreads();  // the previous chunk has been consumed
#pragma unroll
        for (int k = 0; k < 3 * BIG_DC / 4; ++k) {
            const int rowid = w + 4 * k;            // wave-uniform: part = rowid / 16 (X low, X high, Y), dimension = rowid % 16
            const int dd = rowid & (BIG_DC - 1), part = rowid >> 4;
            if (dd < dc) {
                const double ie = p.inv_ell[d0 + dd];
                if (part == 0) sX[dd][lane] = __dmul_rn(X[(size_t)xr0 + (size_t)(d0 + dd) * ldx], ie);
                else if (part == 1) sX[dd][64 + lane] = __dmul_rn(X[(size_t)xr1 + (size_t)(d0 + dd) * ldx], ie);
                else sY[dd][lane] = __dmul_rn(Y[(size_t)yc + (size_t)(d0 + dd) * ldy], ie);
            }
        }
        __syncthreads();
        for (int dd = 0; dd < dc; ++dd) {
            const double2 xa = *reinterpret_cast<const double2 *>(&sX[dd][2 * tx]);
            const double2 xb = *reinterpret_cast<const double2 *>(&sX[dd][64 + 2 * tx]);
            const double xv[4] = {xa.x, xa.y, xb.x, xb.y};
            const double2 *yp = reinterpret_cast<const double2 *>(&sY[dd][ty * 8]);
#pragma unroll
            for (int q2 = 0; q2 < 4; ++q2) {
                const double2 yv = yp[q2];
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    const double e0 = __dsub_rn(xv[a], yv.x), e1 = __dsub_rn(xv[a], yv.y);
                    acc[a][2 * q2] = fma(e0, e0, acc[a][2 * q2]);
                    acc[a][2 * q2 + 1] = fma(e1, e1, acc[a][2 * q2 + 1]);
                }
            }
        }
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int c = col0 + ty * 8 + q;
        if (c >= m) break;
        double v[4];
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const int ra = r + (a >> 1) * 64 + (a & 1);
            v[a] = p.a2 * exp_nonpos(-0.5 * acc[a][q], ec);
            if (same && ra == c) v[a] = p.a2 + diag_add;
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int rh = r + 64 * h;
            const bool w0 = (rh < n) && (!lower || rh >= c), w1 = (rh + 1 < n) && (!lower || rh + 1 >= c);
            double *dst = K + (size_t)rh + (size_t)c * ldk;
            if (vec & 2) store_pair<true>(dst, v[2 * h], v[2 * h + 1], w0, w1, true);
            else store_pair<false>(dst, v[2 * h], v[2 * h + 1], w0, w1, vec != 0);
        }
    }
}

__global__ __launch_bounds__(256) void k_deriv_cov(int kind, const double *__restrict__ x, int n,
                                                   const double *__restrict__ y, int m, double a2,
                                                   double l2, int compat, int lower,
                                                   double *__restrict__ K, size_t ldk, int vec)
{
    const int row0 = blockIdx.x * TILE, col0 = blockIdx.y * TILE;
    if (lower && col0 > row0 + TILE - 1) return;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int r = row0 + 2 * tx;
    const bool ok0 = r < n, ok1 = r + 1 < n;
    const double x0 = ok0 ? x[r] : 0.0, x1 = ok1 ? x[r + 1] : 0.0;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int c = col0 + ty * 8 + q;
        if (c >= m) break;
        const double yv = y[c];
        const double v0 = deriv_cov_val(kind, compat, a2, x0, yv, l2), v1 = deriv_cov_val(kind, compat, a2, x1, yv, l2);
        const bool w0 = ok0 && (!lower || r >= c), w1 = ok1 && (!lower || r + 1 >= c);
        store_pair(K + (size_t)r + (size_t)c * ldk, v0, v1, w0, w1, vec != 0);
    }
}

__global__ __launch_bounds__(256) void k_deriv_elem(int kind, const double *__restrict__ tj,
                                                    const double *__restrict__ tk, size_t len,
                                                    double l2, double *__restrict__ out)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < len;
         i += (size_t)gridDim.x * blockDim.x)
        out[i] = deriv_val(kind, tj[i], tk[i], l2);
}

// Joint covariance of order 2n, rows/cols ordered [values; derivatives]
// (R/ode_gp_library.R:29-30): one exp per point pair feeds QQ, QR, RQ, RR.
__global__ __launch_bounds__(256) void k_joint_cov(const double *__restrict__ t, int n, double a2,
                                                   double l2, double s2, double jitter, int compat,
                                                   int lower, double *__restrict__ K, size_t ldk, int vec, ExpC ec)
{
    const int row0 = blockIdx.x * TILE, col0 = blockIdx.y * TILE;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int r = row0 + 2 * tx;
    const bool ok0 = r < n, ok1 = r + 1 < n;
    const double x0 = ok0 ? t[r] : 0.0, x1 = ok1 ? t[r + 1] : 0.0;
    const double il2 = 1.0 / l2;
    const double a2rr = compat ? 1.0 : a2;  // R/kernels.R:31 precedence bug when compat
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int c = col0 + ty * 8 + q;
        if (c >= n) break;
        const double yv = t[c];
        const double r0 = x0 - yv, r1 = x1 - yv;
        // same argument as derivative_kernels.R:39-41; the lean exp for x <= 0 (<= 1 ulp)
        const double e0 = exp_nonpos(-(r0 * r0 / (2 * l2)), ec), e1 = exp_nonpos(-(r1 * r1 / (2 * l2)), ec);
        double qq0 = a2 * e0, qq1 = a2 * e1;
        const double qr0 = a2 * e0 * r0 * il2, qr1 = a2 * e1 * r1 * il2;
        double rr0 = a2 * e0 * il2 - a2rr * e0 * r0 * r0 * il2 * il2;
        double rr1 = a2 * e1 * il2 - a2rr * e1 * r1 * r1 * il2 * il2;
        if (r == c) { qq0 += s2 + jitter; rr0 += jitter; }
        if (r + 1 == c) { qq1 += s2 + jitter; rr1 += jitter; }
        const bool lo0 = !lower || r >= c, lo1 = !lower || r + 1 >= c;
        // QQ block (rows 0..n, cols 0..n) and RR block (rows n.., cols n..): lower-filtered
        store_pair(K + (size_t)r + (size_t)c * ldk, qq0, qq1, ok0 && lo0, ok1 && lo1, vec != 0);
        store_pair(K + (size_t)(n + r) + (size_t)(n + c) * ldk, rr0, rr1, ok0 && lo0, ok1 && lo1,
                   vec != 0 && (n % 2 == 0));
        // RQ block, rows n.., cols 0..n: t(UD)[i,j] = QR(t_j, t_i) = -qr; entirely below the diagonal
        store_pair(K + (size_t)(n + r) + (size_t)c * ldk, -qr0, -qr1, ok0, ok1, vec != 0 && (n % 2 == 0));
        // QR block, rows 0..n, cols n..: entirely above the diagonal
        if (!lower) store_pair(K + (size_t)r + (size_t)(n + c) * ldk, qr0, qr1, ok0, ok1, vec != 0);
    }
}

__global__ void k_set_row(double *W, size_t ld, int row, const double *src, int n, int ntotal)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < ntotal) W[(size_t)row + (size_t)j * ld] = (j < n) ? src[j] : 0.0;
}

// mode 0: plain copy; 1: lower triangle, strict upper zeroed; 2: lower triangle mirrored (symmetric out)
__global__ __launch_bounds__(256) void k_copy_matrix(const double *__restrict__ src, size_t lds,
                                                     double *__restrict__ dst, size_t ldd, int rows,
                                                     int cols, int mode)
{
    const int r = blockIdx.x * 64 + (threadIdx.x & 63);
    const int c0 = blockIdx.y * 16 + (threadIdx.x >> 6) * 4;
    if (r >= rows) return;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int c = c0 + q;
        if (c >= cols) break;
        double v;
        if (mode == 0) v = src[(size_t)r + (size_t)c * lds];
        else if (mode == 1) v = (r >= c) ? src[(size_t)r + (size_t)c * lds] : 0.0;
        else v = (r >= c) ? src[(size_t)r + (size_t)c * lds] : src[(size_t)c + (size_t)r * lds];
        dst[(size_t)r + (size_t)c * ldd] = v;
    }
}

__global__ void k_add_diag(double *A, size_t ld, int n, double v)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) A[(size_t)i * (ld + 1)] += v;
}

__global__ __launch_bounds__(256) void k_transpose(const double *__restrict__ src, size_t lds,
                                                   double *__restrict__ dst, size_t ldd, int rows, int cols)
{
    __shared__ double tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    const int r0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    for (int q = ty; q < 32; q += 8) {
        const int r = r0 + tx, c = c0 + q;
        tile[q][tx] = (r < rows && c < cols) ? src[(size_t)r + (size_t)c * lds] : 0.0;
    }
    __syncthreads();
    for (int q = ty; q < 32; q += 8) {
        const int c = c0 + tx, r = r0 + q;  // dst[c, r] = src[r, c]
        if (r < rows && c < cols) dst[(size_t)c + (size_t)r * ldd] = tile[tx][q];
    }
}

// B symmetric n x n -> Phi(B)^T stored as the upper triangle: keep i < j, halve i == j, zero i > j
__global__ __launch_bounds__(256) void k_phi_mask(double *B, size_t ld, int n)
{
    const int r = blockIdx.x * 64 + (threadIdx.x & 63);
    const int c0 = blockIdx.y * 16 + (threadIdx.x >> 6) * 4;
    if (r >= n) return;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int c = c0 + q;
        if (c >= n) break;
        double *p = B + (size_t)r + (size_t)c * ld;
        if (r > c) *p = 0.0;
        else if (r == c) *p = 0.5 * *p;
    }
}

inline bool vec_ok(const void *p, size_t ld) { return ((uintptr_t)p % 16 == 0) && (ld % 2 == 0); }


// ---------------------------------------------------------------------------
// Cholesky-factor interpolation (covariance.cpp:49-96, cubic_interpolated_gp.hpp:38-73):
// cubic Hermite blend of two factors and their length-scale tangents, element by element, in
// the reference's operation order.  HBM-bound: 4 matrices read (lower triangles).
// ---------------------------------------------------------------------------
__device__ __forceinline__ double hermite(double y1, double y2, double k1, double k2, double dx, double t)
{
    const double a = k1 * dx - (y2 - y1);
    const double b = -k2 * dx + (y2 - y1);
    return (1 - t) * y1 + t * y2 + t * (1 - t) * (a * (1 - t) + b * t);
}

__global__ __launch_bounds__(256) void k_hermite_blend(const double *__restrict__ L1, const double *__restrict__ L2,
                                                       const double *__restrict__ D1, const double *__restrict__ D2,
                                                       size_t ld, int n, double dx, double t,
                                                       double *__restrict__ out, size_t ldo)
{
    const int i = blockIdx.x * 64 + (threadIdx.x & 63);
    const int j0 = blockIdx.y * 16 + (threadIdx.x >> 6) * 4;
    if (i >= n) return;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int j = j0 + q;
        if (j >= n) break;
        const size_t o = (size_t)i + (size_t)j * ld;
        out[(size_t)i + (size_t)j * ldo] = (j <= i) ? hermite(L1[o], L2[o], D1[o], D2[o], dx, t) : 0.0;
    }
}

// f = blend(l) z without storing the blend: thread = row, blockIdx.y = chunk of HMV_CW columns;
// partial row sums go to part[chunk][row] and are added in ascending chunk order by
// k_hermite_mv_sum (fixed order: results do not depend on scheduling).
// GRAD: the same pass also accumulates (dv/dl) z -- the partial Stan's reverse mode attaches to
// approx_Lz (build_output's `var` overload, models/cubic_interpolated_gp.hpp:6-32; dvdl as written at
// :67) -- into partg: no extra HBM traffic, the four triangles are read once.
constexpr int HMV_CW = 128;
constexpr int HMV_SMALL_N = GPMI_HMV_SMALL_N;  // up to here blend + mat-vec is one launch (k_hermite_mv_small)
template <bool GRAD>
__global__ __launch_bounds__(256) void k_hermite_mv(const double *__restrict__ L1, const double *__restrict__ L2,
                                                    const double *__restrict__ D1, const double *__restrict__ D2,
                                                    size_t ld, int n, double dx, double t, double dtdl,
                                                    const double *__restrict__ z, double *__restrict__ part,
                                                    double *__restrict__ partg)
{
    __shared__ double sz[HMV_CW];
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int j0 = blockIdx.y * HMV_CW;
    if (j0 > blockIdx.x * 256 + 255) return;  // chunk entirely right of this row block's diagonal: stays zero (part is pre-zeroed)
    if (threadIdx.x < HMV_CW) sz[threadIdx.x] = (j0 + threadIdx.x < n) ? z[j0 + threadIdx.x] : 0.0;
    __syncthreads();
    if (i >= n) return;
    const int jend = (i + 1 < j0 + HMV_CW) ? i + 1 : j0 + HMV_CW;  // columns j <= i
    double acc = 0.0, accg = 0.0;
    for (int j = j0; j < jend; ++j) {
        const size_t o = (size_t)i + (size_t)j * ld;
        const double y1 = L1[o], y2 = L2[o], k1 = D1[o], k2 = D2[o];
        acc += hermite(y1, y2, k1, k2, dx, t) * sz[j - j0];
        if constexpr (GRAD) {
            const double a = k1 * dx - (y2 - y1);
            const double b = -k2 * dx + (y2 - y1);
            accg += ((b * (2 - 3 * t) * t + a * (1 + t * (-4 + 3 * t)) - y1 + y2) * dtdl) * sz[j - j0];
        }
    }
    part[(size_t)blockIdx.y * n + i] = acc;
    if constexpr (GRAD) partg[(size_t)blockIdx.y * n + i] = accg;
}

// The same blend + mat-vec for small n in ONE launch (the per-iteration call of the interpolated model at the reference's
// N = 100): thread = row, z staged in LDS by the workgroup (z may be host-mapped: read once), the row's 128-column chunk sums
// formed and added in chunk order -- the additions of k_hermite_mv + k_hermite_mv_sum in the same order (bit-identical) --
// with eight columns (32 loads) in flight per round trip.
template <bool GRAD>
__global__ __launch_bounds__(256) void k_hermite_mv_small(const double *__restrict__ L1, const double *__restrict__ L2,
                                                          const double *__restrict__ D1, const double *__restrict__ D2,
                                                          size_t ld, int n, double dx, double t, double dtdl,
                                                          const double *__restrict__ z, double *__restrict__ f,
                                                          double *__restrict__ fg)
{
    extern __shared__ double szs[];
    for (int j = threadIdx.x; j < n; j += 256) szs[j] = z[j];
    __syncthreads();
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    double tot = 0.0, totg = 0.0;
    for (int j0 = 0; j0 <= i; j0 += HMV_CW) {
        const int jend = (i + 1 < j0 + HMV_CW) ? i + 1 : j0 + HMV_CW;
        double acc = 0.0, accg = 0.0;
        for (int jb = j0; jb < jend; jb += 8) {
            double y1[8], y2[8], k1[8], k2[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int j = jb + q < jend ? jb + q : jend - 1;
                const size_t o = (size_t)i + (size_t)j * ld;
                y1[q] = L1[o];
                y2[q] = L2[o];
                k1[q] = D1[o];
                k2[q] = D2[o];
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                if (jb + q < jend) {
                    acc += hermite(y1[q], y2[q], k1[q], k2[q], dx, t) * szs[jb + q];
                    if constexpr (GRAD) {
                        const double a = k1[q] * dx - (y2[q] - y1[q]);
                        const double b = -k2[q] * dx + (y2[q] - y1[q]);
                        accg += ((b * (2 - 3 * t) * t + a * (1 + t * (-4 + 3 * t)) - y1[q] + y2[q]) * dtdl) * szs[jb + q];
                    }
                }
            }
        }
        tot += acc;
        if constexpr (GRAD) totg += accg;
    }
    f[i] = tot;
    if constexpr (GRAD) fg[i] = totg;
}

__global__ void k_hermite_mv_sum(const double *__restrict__ part, int n, int nchunks, double *__restrict__ f)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    double acc = 0.0;
    const int cmax = i / HMV_CW;  // chunks right of the diagonal hold nothing for this row
    for (int c = 0; c <= cmax && c < nchunks; ++c) acc += part[(size_t)c * n + i];
    f[i] = acc;
}
}  // namespace

void launch_se_cov(const gpmi_ctx *c, hipStream_t s, const double *dX, int n, int ldx, const double *dY, int m, int ldy,
                   const SeParams &p, double diag_add, int lower, double *dK, size_t ldk)
{
    if (n <= 0 || m <= 0) return;
    const int same = (dY == nullptr);
    if (same) { dY = dX; ldy = ldx; }
    dim3 grid((n + BIG_TR - 1) / BIG_TR, (m + BIG_TC - 1) / BIG_TC);                      // k_se_cov_big
    if (p.D <= 3) grid = dim3((n + SE_TR - 1) / SE_TR, (m + SE_TC - 1) / SE_TC);  // k_se_cov<D>
    const int vec = vec_ok(dK, ldk) ? (c->tune.se_nt ? 3 : 1) : 0;  // bit 1: non-temporal stores in k_se_cov<>
    switch (p.D) {
    case 1: hipLaunchKernelGGL(k_se_cov<1>, grid, 256, 0, s, dX, n, ldx, dY, m, ldy, p, diag_add, same, lower, dK, ldk, vec, h_exp); break;
    case 2: hipLaunchKernelGGL(k_se_cov<2>, grid, 256, 0, s, dX, n, ldx, dY, m, ldy, p, diag_add, same, lower, dK, ldk, vec, h_exp); break;
    case 3: hipLaunchKernelGGL(k_se_cov<3>, grid, 256, 0, s, dX, n, ldx, dY, m, ldy, p, diag_add, same, lower, dK, ldk, vec, h_exp); break;
    default:  // measured at N = 16384: D = 8 0.45 ms through the register-resident generic form, 0.30 ms (D = 9) through this one
        hipLaunchKernelGGL(k_se_cov_big, grid, 256, 0, s, dX, n, ldx, dY, m, ldy, p, diag_add, same, lower, dK, ldk, vec, h_exp);
        break;
    }
}

void launch_deriv_cov(hipStream_t s, int kind, const double *dx, int n, const double *dy, int m,
                      double a2, double l, int compat, int lower, double *dK, size_t ldk)
{
    if (n <= 0 || m <= 0) return;
    dim3 grid((n + TILE - 1) / TILE, (m + TILE - 1) / TILE);
    hipLaunchKernelGGL(k_deriv_cov, grid, 256, 0, s, kind, dx, n, dy, m, a2, l * l, compat, lower, dK, ldk,
                       (int)vec_ok(dK, ldk));
}

void launch_deriv_elem(hipStream_t s, int kind, const double *tj, const double *tk, size_t len,
                       double l, double *out)
{
    if (len == 0) return;
    size_t blocks = (len + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(k_deriv_elem, dim3((unsigned)blocks), 256, 0, s, kind, tj, tk, len, l * l, out);
}

void launch_joint_cov(hipStream_t s, const double *dt, int n, double a2, double l, double s2,
                      double jitter, int compat, int lower, double *dK, size_t ldk)
{
    if (n <= 0) return;
    dim3 grid((n + TILE - 1) / TILE, (n + TILE - 1) / TILE);
    hipLaunchKernelGGL(k_joint_cov, grid, 256, 0, s, dt, n, a2, l * l, s2, jitter, compat, lower, dK, ldk,
                       (int)vec_ok(dK, ldk), h_exp);
}

void launch_set_row(hipStream_t s, double *W, size_t ld, int row, const double *src, int n, int ntotal)
{
    if (ntotal <= 0) return;
    hipLaunchKernelGGL(k_set_row, dim3((ntotal + 255) / 256), 256, 0, s, W, ld, row, src, n, ntotal);
}

void launch_copy_matrix(hipStream_t s, const double *src, size_t lds, double *dst, size_t ldd,
                        int rows, int cols, int mode)
{
    if (rows <= 0 || cols <= 0) return;
    dim3 grid((rows + 63) / 64, (cols + 15) / 16);
    hipLaunchKernelGGL(k_copy_matrix, grid, 256, 0, s, src, lds, dst, ldd, rows, cols, mode);
}

void launch_add_diag(hipStream_t s, double *A, size_t ld, int n, double v)
{
    if (n <= 0) return;
    hipLaunchKernelGGL(k_add_diag, dim3((n + 255) / 256), 256, 0, s, A, ld, n, v);
}

void launch_transpose(hipStream_t s, const double *src, size_t lds, double *dst, size_t ldd, int rows, int cols)
{
    if (rows <= 0 || cols <= 0) return;
    dim3 grid((rows + 31) / 32, (cols + 31) / 32);
    hipLaunchKernelGGL(k_transpose, grid, 256, 0, s, src, lds, dst, ldd, rows, cols);
}

void launch_phi_mask(hipStream_t s, double *B, size_t ld, int n)
{
    if (n <= 0) return;
    dim3 grid((n + 63) / 64, (n + 15) / 16);
    hipLaunchKernelGGL(k_phi_mask, grid, 256, 0, s, B, ld, n);
}

void launch_hermite_blend(hipStream_t s, const double *L1, const double *L2, const double *D1, const double *D2,
                          size_t ld, int n, double x1, double x2, double l, double *out, size_t ldo)
{
    if (n <= 0) return;
    const double t = (l - x1) / (x2 - x1);
    hipLaunchKernelGGL(k_hermite_blend, dim3((n + 63) / 64, (n + 15) / 16), 256, 0, s, L1, L2, D1, D2, ld, n, x2 - x1, t,
                       out, ldo);
}

int hermite_mv_chunks(int n) { return (n + HMV_CW - 1) / HMV_CW; }

// part: 2 * hermite_mv_chunks(n) * n doubles, pre-zeroed; dfdl == nullptr: value only
void launch_hermite_mv(hipStream_t s, const double *L1, const double *L2, const double *D1, const double *D2,
                       size_t ld, int n, double x1, double x2, double l, const double *z, double *part, double *f,
                       double *dfdl)
{
    if (n <= 0) return;
    const double t = (l - x1) / (x2 - x1);
    const double dtdl = 1 / (x2 - x1);
    if (n <= HMV_SMALL_N) {  // one launch; `part` is not used
        const dim3 g1((n + 255) / 256);
        if (dfdl)
            hipLaunchKernelGGL(k_hermite_mv_small<true>, g1, 256, (size_t)n * sizeof(double), s, L1, L2, D1, D2, ld, n, x2 - x1, t, dtdl, z, f, dfdl);
        else
            hipLaunchKernelGGL(k_hermite_mv_small<false>, g1, 256, (size_t)n * sizeof(double), s, L1, L2, D1, D2, ld, n, x2 - x1, t, dtdl, z, f, dfdl);
        return;
    }
    const int nch = hermite_mv_chunks(n);
    double *partg = part + (size_t)nch * n;
    const dim3 grid((n + 255) / 256, nch);
    if (dfdl)
        hipLaunchKernelGGL(k_hermite_mv<true>, grid, 256, 0, s, L1, L2, D1, D2, ld, n, x2 - x1, t, dtdl, z, part, partg);
    else
        hipLaunchKernelGGL(k_hermite_mv<false>, grid, 256, 0, s, L1, L2, D1, D2, ld, n, x2 - x1, t, dtdl, z, part, partg);
    hipLaunchKernelGGL(k_hermite_mv_sum, dim3((n + 255) / 256), 256, 0, s, part, n, nch, f);
    if (dfdl) hipLaunchKernelGGL(k_hermite_mv_sum, dim3((n + 255) / 256), 256, 0, s, partg, n, nch, dfdl);
}
