// Internal declarations shared by the libgpmi translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/gpmi.h"

#define GPMI_NB 128          // inner panel width == diagonal-block order
#define GPMI_FPACK 9216      // doubles in one packed panel-factor buffer (36 tiles x 256)
#define GPMI_FPACK_SLOTS 16   // per-context ring of such buffers: one per 128-column panel of an outer block
#define GPMI_MAXD 8          // dimensions the register-resident fast path handles
#define GPMI_MAXD_BIG 64     // dimensions the generic path handles (inverse length-scales in kernel arguments)
#define GPMI_SMALL_PTS 128   // grid points per launch of the one-workgroup-per-point small-N kernel
#define GPMI_SMALL_PTS_ARD 32 // ... with a length-scale per dimension and point

typedef double d4 __attribute__((ext_vector_type(4)));

struct SeParams {            // squared-exponential hyper-parameters, kernel-argument resident
    double a2;               // alpha^2
    double inv_ell[GPMI_MAXD_BIG];
    int D;
};

// Algorithm switches of one context (gpmi_set_option).  Per context, never process-global: two
// contexts or two host threads do not change each other's algorithm; grid lanes copy their root's.
struct gpmi_tuning {
    int syrk_order;       // tile walk of multi-round trailing updates: 0 row-major, 2 XCD-partitioned bands (1: probe build, padded super-tiles)
    int stagger;          // (mode << 16) | number of s_sleep(127) (~3.5 us each) for the late workgroup of a CU pair
    int fuse_diag;        // default 15; bit 0: in-block GEMMs, bit 1: trailing SYRK (multi-round), bit 2: sub-tiled diagonal tile, bit 3: single-round SYRK (sub-tiled)
    int diag_waves;       // (probe build only) 4: k_potrf_diag4 (default), 5: k_potrf_diag
    int nb_adapt;         // outer-block width re-chosen per block from the order of the matrix still to update
    int nb_thr[3];        // ... >= nb_thr[0]: 1024 columns, >= nb_thr[1]: 512, >= nb_thr[2]: 256, below: 128
    int ksplit, ksplit_max;
    int block_recursive;
    int se_nt;            // non-temporal stores in k_se_cov<>
    int gemm_variant, rect_auto;  // A/B kernels: honoured by the probe build (-DGPMI_PROBES) only
    int small_n;          // grids of marginal likelihoods at n <= small_n (and D <= GPMI_MAXD): one workgroup per point, one launch (0: off)
    int small_n1;         // ... a single evaluation (or a grid of fewer than 6 points) up to this n: beyond it the multi-CU launch chain is faster
    int small_m;          // partial factorisation of <= small_m rows: one workgroup, one launch (0: off)
    int small_ng1, small_ng; // value + gradient by one workgroup: one evaluation up to n <= small_ng1, several (a sampler's chains) up to small_ng (<= 256; 0: off)
    int grad_aug_n, grad_aug_ng;  // gpmi_logml_grad / _grid: K^-1 and K^-1 y from one augmented partial factorisation up to this n (0: off)
    int small_gc;            // gpmi_gp_condition: one workgroup, one launch, up to n + m + 1 <= small_gc rows (0: off)
    int small_sd, small_sdb; // sample_derivs_batch: one workgroup per draw when n + m + 1 <= small_sd rows and at least small_sdb (n + m + 1)^2 / 400^2 draws (0: off)
    int small_n2, small_g2;  // grids of >= small_g2 (n / 1024)^2 + 2 points: one workgroup per point up to n <= small_n2 (every CU a problem of its own)
};
void gpmi_tuning_defaults(gpmi_tuning *t);

struct gpmi_ctx {
    int device;
    int pid;
    gpmi_tuning tune;
    hipStream_t own_stream;
    hipStream_t stream;      // stream in use (own or caller's)
    // factorisation workspace: column-major, leading dimension ld, ncols columns (+ slack)
    double *W;
    size_t W_bytes;
    int ld, ncols;
    double *Fpack;           // packed factors of the current diagonal block
    double *scratch;         // small device scratch (results, staging)
    size_t scratch_bytes;
    int *d_info;
    int *d_ctr;              // zeroed counters: [0..2] persistent trailing update (tile, retired, early exits),
                             // [8] sub-tile counter of the fused in-block GEMM
    int ncu;                 // compute units of the device
    double *d_out;           // 3 doubles
    double *d_fin;           // slice sums of the finalize kernels
    double *h_pin;           // pinned, device-mapped host buffer: inputs and results of small host-buffer calls travel without a copy call
    double *h_pin_dev;       // its device address
    size_t h_pin_bytes;
    double *d_spar;          // device-parameter small-N grids: GPMI_SMALL_PAR doubles + one work int per point
    int *d_sinfo;
    int spar_pts;            // capacity (points)
    int pin_seq;             // sequence number of the completion flag in h_pin (one-launch host-buffer calls)
    // generic device staging buffers for the host-pointer API
    double *stage[4];
    size_t stage_bytes[4];
    int nb_outer;            // outer panel width (multiple of GPMI_NB)
    int lookahead;           // (probe build only) two-stream look-ahead over outer blocks: 0 off (default), 1 on
    hipStream_t pstream;     // panel stream of the look-ahead when no calibrated pair exists
    hipEvent_t evM;
    hipEvent_t evP, evU;     // panel done / next-panel columns updated
    int timing;
    hipEvent_t ev[4];
    double last_ms[3];
    // per-kernel HIP-event timing (bench): category 0 covariance build, 1 trailing SYRK,
    // 2 panel kernels (diag potrf + panel solve + in-block update)
    int ktiming;
    void *ktimer;            // KTimer*
    // grid lanes: extra internal contexts (own workspace + streams) so that independent grid
    // points overlap -- one point's panel phase and SYRK tails run under another's bulk update
    int grid_lanes;          // 0 = auto
    int lane_lookahead;      // panel look-ahead inside each lane of a multi-lane grid (default off)
    // Cholesky-factor interpolation table (gpmi_interp_*): P stacked n x itp_ld matrices each
    double *itp_L, *itp_dL, *itp_part;
    double *itp_lp;          // host copy of the P length-scales
    int itp_P, itp_n;
    size_t itp_ld;
    gpmi_ctx *lane[7];
    hipEvent_t evFork, evJoin;
    // Dispatch streams on distinct command-processor pipes (root context), found by probing:
    // kernels of two queues overlap freely only if the queues sit on different pipes
    // (see calibrate_streams in gpmi_api.hip).
    hipStream_t qstream[4];
    int nq;                  // number of mutually concurrent streams found (0 = not probed yet)
    int cal_want;            // ... the largest number a calibration run has searched for so far
    int lanes_active;        // inside a multi-lane call (grid, table build, batch): other lanes' kernels share the chip
    int calibrate;           // 1: run grid lanes on qstream[]; 0: on the lanes' plain streams
};

// event-pair recorder; begin/end bracket one launch on the context's stream
void kt_begin(gpmi_ctx *c, int cat, hipStream_t s = nullptr);
void kt_end(gpmi_ctx *c, int cat, double work, hipStream_t s = nullptr, int flag = 0);
int gpmi_lookahead_streams(gpmi_ctx *c);  // two concurrently dispatching streams for the look-ahead (calibrated, or a panel stream)

// ---- error plumbing -------------------------------------------------------
int gpmi_fail(int code, const char *fmt, ...);
#define HIPCHK(expr)                                                                   \
    do {                                                                               \
        hipError_t e_ = (expr);                                                        \
        if (e_ != hipSuccess)                                                          \
            return gpmi_fail(GPMI_EHIP, "%s failed: %s (%s:%d)", #expr,               \
                             hipGetErrorString(e_), __FILE__, __LINE__);               \
    } while (0)

// ---- kernel launchers (se_kernels.hip) -------------------------------------
// K (n x m, ldk) from device points; Y == X when dY == nullptr.
void launch_se_cov(const gpmi_ctx *c, hipStream_t s, const double *dX, int n, int ldx, const double *dY, int m, int ldy,
                   const SeParams &p, double diag_add, int lower, double *dK, size_t ldk);
void launch_deriv_cov(hipStream_t s, int kind, const double *dx, int n, const double *dy, int m,
                      double a2, double l, int compat, int lower, double *dK, size_t ldk);
void launch_deriv_elem(hipStream_t s, int kind, const double *tj, const double *tk, size_t len,
                       double l, double *out);
// joint 2n x 2n [[QQ+s2 I, QR],[RQ, RR]] + jitter I; lower != 0 writes i >= j only
void launch_joint_cov(hipStream_t s, const double *dt, int n, double a2, double l, double s2,
                      double jitter, int compat, int lower, double *dK, size_t ldk);
void launch_set_row(hipStream_t s, double *W, size_t ld, int row, const double *src, int n,
                    int ntotal);  // W[row, j] = j < n ? src[j] : 0, j < ntotal
void launch_copy_matrix(hipStream_t s, const double *src, size_t lds, double *dst, size_t ldd,
                        int rows, int cols, int mode);  // mode 0 full, 1 lower (upper zero), 2 lower->symmetric
void launch_add_diag(hipStream_t s, double *A, size_t ld, int n, double v);
void launch_transpose(hipStream_t s, const double *src, size_t lds, double *dst, size_t ldd, int rows, int cols);
void launch_hermite_blend(hipStream_t s, const double *L1, const double *L2, const double *D1, const double *D2,
                          size_t ld, int n, double x1, double x2, double l, double *out, size_t ldo);
void launch_hermite_mv(hipStream_t s, const double *L1, const double *L2, const double *D1, const double *D2,
                       size_t ld, int n, double x1, double x2, double l, const double *z, double *part, double *f,
                       double *dfdl /* nullable: (dv/dl) z, the reverse-mode partial of approx_Lz */);
int hermite_mv_chunks(int n);
#define GPMI_HMV_SMALL_N 256    // approx_Lz up to this n: one launch, z read straight from (possibly host-mapped) memory
void launch_phi_mask(hipStream_t s, double *B, size_t ld, int n); // keep upper, halve diag, zero strict lower

// ---- kernel launchers (chol_kernels.hip) -----------------------------------
// Factor the first nfac columns of the M x ncol lower-stored matrix in W (right-looking,
// blocked); rows nfac..M-1 become L21 / the Schur complement of the trailing block.
int launch_potrf_partial(gpmi_ctx *c, double *W, size_t ld, int M, int ncol, int nfac, int *d_info,
                         double *Fpack_all /* nullable: keep every block's packed factors */);
// X <- X * L^-T for the rows [row0, M) of columns [0, n) using packed factors saved by a
// previous launch_potrf_partial(..., Fpack_all)
int launch_trsm_right(gpmi_ctx *c, const double *L, size_t ldl, int n, double *X, size_t ldx,
                      int mrows, const double *Fpack_all, int upper_tri = 0 /* 1: X upper triangular, 2: lower triangle of the result only */);
// C (lower tiles only) = A B^T for lower-triangular A and B the transpose of a lower-triangular matrix (n^3 / 3 flops)
void launch_gemm_tri_lower(hipStream_t s, const double *A, size_t lda, const double *B, size_t ldb, double *C, size_t ldc, int n);
// C (M x N) = beta_is_one ? C - A B^T : A B^T   (A: M x K, B: N x K, column-major)
void launch_gemm_nt(const gpmi_ctx *c, hipStream_t s, const double *A, size_t lda, const double *B, size_t ldb,
                    double *C, size_t ldc, int M, int N, int K, int accumulate_minus);
void launch_syrk_uut(const gpmi_ctx *c, hipStream_t s, const double *U, size_t ldu, double *C, size_t ldc, int n);
void launch_pack_factors(hipStream_t s, const double *L, size_t ldl, int n, double *Fpack_all);
// small-N marginal likelihood: build + factorisation + solve + log-det by ONE workgroup per point
void small_ws_layout(int n, size_t *ld, size_t *stride);
void launch_logml_small(hipStream_t s, const double *dX, int n, int ldx, const double *dy, const SeParams &p, double diag_add,
                        double *W, size_t ld, double *d_out3, int *d_info_out, int *d_info_work,
                        double *stage /* nullable: device staging for host-mapped X, y */,
                        int *done = nullptr /* nullable: host-mapped completion flag, set to seq at the end */, int seq = 0);
void launch_logml_small_batch_ard(hipStream_t s, const double *dX, int n, int ldx, int D, const double *dy, const double *alpha,
                                  const double *ell /* G x D, point-major */, const double *sigma, int G, double jitter,
                                  double *Wall, double *d_out3, int *d_info_out, int *d_info_work);
void launch_logml_small_batch(hipStream_t s, const double *dX, int n, int ldx, int D, const double *dy, const double *alpha,
                              const double *rho, const double *sigma, int G, double jitter, double *Wall, double *d_out3,
                              int *d_info_out, int *d_info_work);
// value + gradient sums (GPMI_SMALL_GRAD_RES doubles per point: logml, sum log L_ii, z'z, then the GRAD_NS = 10 contraction
// sums) by one workgroup per point, n <= 256; the workspace holds TWO slices of small_ws_layout per point (W, then U = L^-T)
#define GPMI_SMALL_GRAD_RES (3 + 2 + GPMI_MAXD)
void launch_logml_grad_small(hipStream_t s, const double *dX, int n, int ldx, const double *dy, const SeParams &p, double diag_add,
                             double *W, double *d_res, int *d_info_out, int *d_info_work, double *stage, int *done = nullptr, int seq = 0);
void launch_logml_grad_small_batch(hipStream_t s, const double *dX, int n, int ldx, int D, const double *dy, const double *alpha,
                                   const double *rho, const double *sigma, int G, double jitter, double *Wall, double *d_res,
                                   int *d_info_out, int *d_info_work, double *stage = nullptr /* G n (D + 1) doubles: X, y host-mapped */,
                                   int *done = nullptr, int seq = 0, int *arrive = nullptr /* zeroed device int, left zero */);
// B posterior draws of the derivative process, one workgroup each (workspace: B slices of small_ws_layout(n + m); d_par: 3 B doubles,
// d_info_work: 2 B ints)
void launch_sample_derivs_small_batch(hipStream_t s, const double *dt, int n, const double *dts, int m, const double *dY,
                                      const double *params, int B, double jitter, const double *dZ, double *d_par, double *Wall,
                                      double *d_draws, double *d_mus, int *d_status, int *d_info_work);
// gp_condition by one workgroup (workspace: one slice of small_ws_layout(n + m)); t, ts, y / Kn, mn, info_out may be host-mapped (stage != null)
void launch_gp_condition_small(hipStream_t s, const double *t, int n, const double *ts, int m, const double *y, int kindK, int kindS,
                               int kindSS, int compat, double a2, double l2, double s2, double jitter, double *W, double *Kn, size_t ldo,
                               double *mn, int *info_out, int *d_info_work, double *stage, int *done = nullptr, int seq = 0);
// f = chol(K + diag_add I) z by one workgroup (n <= 256, D <= GPMI_MAXD); X, z / f, info_out may be host-mapped (stage != null)
void launch_exact_gp_small(hipStream_t s, const double *X, int n, int ldx, const double *z, const SeParams &p, double diag_add,
                           double *W, double *f, int *info_out, int *d_info_work, double *stage, int *done = nullptr, int seq = 0);
// rbf_cov_chol (L and dL/dl) for P <= 64 length-scales, one workgroup each, n <= 128 (workspace: 3 P slices of small_ws_layout(n));
// x / Lout, dLout, info_out may be host-mapped (stage != null: P n doubles of device scratch)
void launch_rbf_cov_chol_small(hipStream_t s, const double *x, int n, const double *ls, int P, double *Wall, double *Lout, double *dLout,
                               size_t ostride, size_t ldo, int *info_out, int *d_info_work, double *stage);
// any number of points, parameters uploaded to d_par (G * GPMI_SMALL_PAR doubles) in stream order; ell: one per point (n_ell == 1) or D per point
#define GPMI_SMALL_PAR (2 + GPMI_MAXD)
#define GPMI_SMALL_NMAX 1024   // n * D <= 9216: the scaled coordinates are staged in the workgroup's LDS
#define GPMI_SMALL_DEV_PTS 512 // points per launch of the device-parameter form (workspace slices held at a time)
void launch_logml_small_batch_dev(hipStream_t s, const double *dX, int n, int ldx, int D, const double *dy, const double *alpha,
                                  const double *ell, int n_ell, const double *sigma, int G, double jitter, double *d_par,
                                  double *Wall, double *d_out3, int *d_info_out, int *d_info_work);
// inverses of L's 128 x 128 diagonal blocks (ceil(n / 128) x 128 x 128 doubles, tmp the same) from packed factors
void launch_diag_inverses(hipStream_t s, const double *Fpack_all, int n, double *Dinv, double *tmp);
// t = L^-1 k for ONE right-hand side in one launch (k_trsv_wave); k and t are different buffers of n doubles
int launch_trsv_lower(hipStream_t s, const double *L, size_t ldl, int n, const double *k, double *t, const double *Dinv,
                      int *ticket /* zeroed device int of the calling context (d_ctr + 32) */);
void launch_get_row(hipStream_t s, const double *W, size_t ld, int row, int col0, int m, double scale, double *out);
void launch_logml_finalize(hipStream_t s, const double *W, size_t ld, int n, int zrow,
                           const int *d_info, double *d_out3, int *d_info_out, double *part /* 2 ceil(n/256) doubles */);
int trmv_lower_chunks(int n);
void launch_trmv_lower(hipStream_t s, const double *L, size_t ldl, int n, const double *z, double *f,
                       double *part /* trmv_lower_chunks(n) * n doubles */);
#ifdef GPMI_PROBES
void launch_syrk_probe(const gpmi_ctx *c, hipStream_t s, const double *P, size_t ldp, double *C, size_t ldc, int m, int k);
void launch_probe_mfma(hipStream_t s, const double *A, const double *B, double *D);
void launch_probe_peak(hipStream_t s, double *sink, int iters, int *blocks, int *threads);
#endif
