// C-ABI implementation of libgpmi (include/gpmi.h).  Host-side orchestration only:
// every arithmetic operation on matrix data runs in a HIP kernel on the context's
// stream.  There is no CPU fallback path.
#include "gpmi_internal.h"

#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include <chrono>
#include <vector>

static thread_local char g_err[512] = "";

int gpmi_fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}

extern "C" const char *gpmi_last_error(void) { return g_err; }
extern "C" int gpmi_version(void) { return GPMI_VERSION; }

// HIP / HSA state does not survive fork() and cannot be re-initialised in the child of a process
// that already initialised it (parallel::mclapply, pendulum_fit.R:268): the first call that touches
// the runtime records the pid, every later one from another pid answers GPMI_EFORK instead of
// reaching into the stale runtime.
static int g_hip_pid = 0;
static int fork_guard(void)
{
    const int pid = (int)getpid();
    if (g_hip_pid == 0) g_hip_pid = pid;
    if (g_hip_pid != pid)
        return gpmi_fail(GPMI_EFORK, "libgpmi initialised the GPU in process %d; this is forked child %d, where HIP cannot be "
                                     "re-initialised: start workers as fresh processes or fork before the first gpmi call",
                         g_hip_pid, pid);
    return 0;
}

extern "C" int gpmi_device_count(int *count)
{
    if (!count) return gpmi_fail(GPMI_EARG, "count is NULL");
    *count = 0;
    int rc = fork_guard();
    if (rc) return rc;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        *count = 0;
        return gpmi_fail(GPMI_ENODEV, "hipGetDeviceCount: %s", hipGetErrorString(e));
    }
    *count = n;
    return 0;
}

// ---- per-kernel event timing --------------------------------------------------
struct KTimer {
    std::vector<hipEvent_t> a[3], b[3];
    std::vector<double> w[3];     // work of each bracket
    std::vector<char> flag[3];    // caller's mark (category 1: launch of more than one round of tiles)
    size_t used[3] = {0, 0, 0};
    double work[3] = {0, 0, 0};
};

void kt_begin(gpmi_ctx *c, int cat, hipStream_t st)
{
    if (!c->ktiming) return;
    KTimer *k = (KTimer *)c->ktimer;
    if (k->used[cat] == k->a[cat].size()) {
        hipEvent_t e0 = nullptr, e1 = nullptr;
        (void)hipEventCreate(&e0);
        (void)hipEventCreate(&e1);
        k->a[cat].push_back(e0);
        k->b[cat].push_back(e1);
        k->w[cat].push_back(0.0);
        k->flag[cat].push_back(0);
    }
    (void)hipEventRecord(k->a[cat][k->used[cat]], st ? st : c->stream);
}

void kt_end(gpmi_ctx *c, int cat, double work, hipStream_t st, int flag)
{
    if (!c->ktiming) return;
    KTimer *k = (KTimer *)c->ktimer;
    (void)hipEventRecord(k->b[cat][k->used[cat]], st ? st : c->stream);
    k->w[cat][k->used[cat]] = work;
    k->flag[cat][k->used[cat]] = (char)flag;
    k->used[cat]++;
    k->work[cat] += work;
}

extern "C" int gpmi_kernel_timing(gpmi_ctx *c, int reset, double *out9);

// ---- stream calibration -------------------------------------------------------
// Kernels of two HIP streams overlap freely only if the streams' hardware queues are served by
// different command-processor pipes (MI355X: 4 per process as observed; queues on one pipe are
// time-sliced in ~56 us quanta, which triples the latency of every small kernel that runs next
// to another queue's trailing update).  Neither HIP nor HSA tells which pipe a queue is on, so
// it is measured: a long many-round kernel on stream a, a one-workgroup kernel on stream b and
// the time until b's kernel is done -- ~15 us if the two queues dispatch concurrently, a time
// slice or the whole long kernel if not.  Candidates are dedicated queues: the CU-mask API gives
// every stream its own hardware queue instead of one shared out of the runtime's pool.
namespace {
__global__ __launch_bounds__(256) void k_cal_long(int *sink, int spin)
{
    __shared__ int pad[18 * 1024];  // 72 KiB: two workgroups per CU, like the trailing update
    pad[threadIdx.x] = threadIdx.x;
    __syncthreads();
    int acc = 0;
    for (int i = 0; i < spin; ++i) {
        __builtin_amdgcn_s_sleep(100);
        acc += pad[(threadIdx.x + i) & 255];
    }
    if (acc == 0x7fffffff) sink[0] = acc;
}
__global__ void k_cal_short(int *sink)
{
    if (sink[1] == 0x7fffffff) sink[2] = 1;
}
}  // namespace

static int streams_concurrent(gpmi_ctx *c, hipStream_t a, hipStream_t b, bool *conc)
{
    int *sink = c->d_info + 8;
    int votes = 0;
    for (int rep = 0; rep < 3; ++rep) {
        HIPCHK(hipDeviceSynchronize());
        const auto t0 = std::chrono::steady_clock::now();
        hipLaunchKernelGGL(k_cal_long, dim3(4096), 256, 0, a, sink, 24);
        hipLaunchKernelGGL(k_cal_short, dim3(1), 64, 0, b, sink);
        HIPCHK(hipStreamSynchronize(b));
        const auto t1 = std::chrono::steady_clock::now();
        HIPCHK(hipDeviceSynchronize());
        const auto t2 = std::chrono::steady_clock::now();
        const double tb = std::chrono::duration<double>(t1 - t0).count(), ta = std::chrono::duration<double>(t2 - t0).count();
        if (getenv("GPMI_DEBUG")) fprintf(stderr, "[gpmi] probe: short %.0f us, long %.0f us\n", 1e6 * tb, 1e6 * ta);
        votes += (tb < 45e-6 && tb < 0.25 * ta);
    }
    *conc = votes >= 2;
    return 0;
}

#ifdef GPMI_PROBES
static int full_mask_stream(gpmi_ctx *c, hipStream_t *out)
{
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, c->device));
    const int words = (prop.multiProcessorCount + 31) / 32;
    std::vector<uint32_t> mask(words, 0xFFFFFFFFu);
    if (prop.multiProcessorCount % 32) mask[words - 1] = (1u << (prop.multiProcessorCount % 32)) - 1u;
    HIPCHK(hipExtStreamCreateWithCUMask(out, (uint32_t)words, mask.data()));
    hipLaunchKernelGGL(k_cal_short, dim3(1), 64, 0, *out, c->d_info + 8);  // binds the hardware queue
    HIPCHK(hipStreamSynchronize(*out));
    return 0;
}
#endif

// up to 4 dedicated streams that are pairwise concurrent (one per pipe)
// want: concurrent streams the caller can use (lanes, at most 4).  The search is resumed when a later call wants
// more than an earlier one found: which hardware queue a new stream lands on depends on the streams that exist at
// that moment (the runtime deals its few hardware queues round-robin), so a calibration run next to ONE lane context
// may see only two distinct pipes where one next to three finds four -- interp_build / grids then ran lanes 2 and 3
// on the streams of lanes 0 and 1, i.e. serialised behind them (tools/interp_build_bench.py: 4 lanes == 2 lanes).
static int calibrate_streams(gpmi_ctx *c, int want = 4)
{
    if (want > 4) want = 4;
    if (c->nq >= want || c->cal_want >= want) return 0;
    c->cal_want = want;
    int nq = c->nq, rc;
    for (int i = 0; i < 10 && nq < want; ++i) {
        hipStream_t cand;
#ifdef GPMI_PROBES
        const char *kind = getenv("GPMI_CAL_KIND");  // experiment: 1 = dedicated (CU-mask API) queues
        if (kind && atoi(kind) == 1) {
            if ((rc = full_mask_stream(c, &cand))) return rc;
        } else
#endif
        {
            HIPCHK(hipStreamCreateWithFlags(&cand, hipStreamNonBlocking));
            hipLaunchKernelGGL(k_cal_short, dim3(1), 64, 0, cand, c->d_info + 8);  // binds the hardware queue
            HIPCHK(hipStreamSynchronize(cand));
        }
        bool ok = true;
        for (int k = 0; k < nq && ok; ++k)
            if ((rc = streams_concurrent(c, c->qstream[k], cand, &ok))) return rc;
        if (ok) c->qstream[nq++] = cand;
        else HIPCHK(hipStreamDestroy(cand));
    }
    c->nq = nq;
    if (getenv("GPMI_DEBUG")) fprintf(stderr, "[gpmi] stream calibration: %d concurrent dispatch streams\n", nq);
    return 0;
}

int gpmi_lookahead_streams(gpmi_ctx *c)
{
    if (c->calibrate && c->nq == 0) {
        int rc = calibrate_streams(c, 2);
        if (rc) return rc;
    }
    if (c->nq >= 2) return 0;
    if (!c->pstream) HIPCHK(hipStreamCreateWithFlags(&c->pstream, hipStreamNonBlocking));
    return 0;
}

// ---- context ---------------------------------------------------------------
static int enter(gpmi_ctx *c)
{
    if (!c) return gpmi_fail(GPMI_EARG, "ctx is NULL");
    if (c->pid != (int)getpid())
        return gpmi_fail(GPMI_EFORK, "gpmi context used from a forked child (pid %d, created in %d); "
                                     "create one context per process", (int)getpid(), c->pid);
    HIPCHK(hipSetDevice(c->device));
    return 0;
}
#define ENTER(c)                 \
    do {                         \
        int rc_ = enter(c);      \
        if (rc_) return rc_;     \
    } while (0)

extern "C" int gpmi_destroy(gpmi_ctx *c);

// HIP call inside gpmi_create: on failure everything allocated so far is released (gpmi_destroy
// accepts a partially built context: every member is null-checked)
#define CREATE_CHK(expr)                                                                          \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess) {                                                                   \
            const int rc_ = gpmi_fail(e_ == hipErrorOutOfMemory ? GPMI_ENOMEM : GPMI_EHIP, "%s failed: %s (%s:%d)", #expr, \
                                      hipGetErrorString(e_), __FILE__, __LINE__);                 \
            gpmi_destroy(c);                                                                      \
            return rc_;                                                                           \
        }                                                                                         \
    } while (0)

extern "C" int gpmi_create(gpmi_ctx **out, int device)
{
    if (!out) return gpmi_fail(GPMI_EARG, "ctx out pointer is NULL");
    *out = nullptr;
    int rc_fork = fork_guard();
    if (rc_fork) return rc_fork;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
        return gpmi_fail(GPMI_ENODEV, "no HIP device visible (libgpmi has no CPU fallback)");
    if (device < 0 || device >= n) return gpmi_fail(GPMI_EARG, "device %d out of range [0,%d)", device, n);
    HIPCHK(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return gpmi_fail(GPMI_ENODEV, "device %d is %s; libgpmi is built for gfx950 only", device, prop.gcnArchName);
    gpmi_ctx *c = (gpmi_ctx *)calloc(1, sizeof(gpmi_ctx));
    if (!c) return gpmi_fail(GPMI_ENOMEM, "host allocation failed");
    c->device = device;
    c->pid = (int)getpid();
    gpmi_tuning_defaults(&c->tune);
    c->nb_outer = 0;  // auto
    c->lookahead = 0; // (probe build: two-stream look-ahead over outer blocks, opt-in)
    c->calibrate = 1;
    c->ncu = prop.multiProcessorCount;
    // test hook: GPMI_FAIL_CREATE_AT=k makes the k-th resource acquisition below fail (leak tests of the unwind path)
    int step = 0;
    const char *fail_env = getenv("GPMI_FAIL_CREATE_AT");
    const int fail_at = fail_env ? atoi(fail_env) : 0;
#define CREATE_STEP(expr) CREATE_CHK((++step == fail_at) ? hipErrorOutOfMemory : (expr))
    CREATE_STEP(hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking));
    c->stream = c->own_stream;
    CREATE_STEP(hipEventCreateWithFlags(&c->evP, hipEventDisableTiming));
    CREATE_STEP(hipEventCreateWithFlags(&c->evU, hipEventDisableTiming));
    CREATE_STEP(hipEventCreateWithFlags(&c->evM, hipEventDisableTiming));
    CREATE_STEP(hipEventCreateWithFlags(&c->evFork, hipEventDisableTiming));
    CREATE_STEP(hipEventCreateWithFlags(&c->evJoin, hipEventDisableTiming));
    CREATE_STEP(hipMalloc((void **)&c->Fpack, (size_t)GPMI_FPACK_SLOTS * GPMI_FPACK * sizeof(double)));
    CREATE_STEP(hipMalloc((void **)&c->d_info, 64));
    CREATE_STEP(hipMalloc((void **)&c->d_ctr, 2048));
    CREATE_STEP(hipMemsetAsync(c->d_ctr, 0, 2048, c->own_stream));
    CREATE_STEP(hipStreamSynchronize(c->own_stream));
    CREATE_STEP(hipMalloc((void **)&c->d_out, 64));
    CREATE_STEP(hipMalloc((void **)&c->d_fin, 65536));  // slice sums of the finalize kernels: 2 doubles per 256 rows
    for (int i = 0; i < 4; ++i) CREATE_STEP(hipEventCreate(&c->ev[i]));
#undef CREATE_STEP
    *out = c;
    return 0;
}

// Releases everything a context owns; every member may be null (partially built context of a failed gpmi_create).
extern "C" int gpmi_destroy(gpmi_ctx *c)
{
    if (!c) return 0;
    if (c->pid == (int)getpid()) {
        (void)hipSetDevice(c->device);
        if (c->stream) (void)hipStreamSynchronize(c->stream);
        for (int l = 0; l < 7; ++l)
            if (c->lane[l]) gpmi_destroy(c->lane[l]);
        double *dev_bufs[] = {c->W, c->Fpack, c->itp_L, c->itp_dL, c->itp_part, c->d_out, c->d_fin, c->scratch,
                              c->stage[0], c->stage[1], c->stage[2], c->stage[3]};
        for (double *b : dev_bufs)
            if (b) (void)hipFree(b);
        if (c->d_info) (void)hipFree(c->d_info);
        if (c->d_ctr) (void)hipFree(c->d_ctr);
        if (c->h_pin) (void)hipHostFree(c->h_pin);
        if (c->d_spar) (void)hipFree(c->d_spar);
        if (c->d_sinfo) (void)hipFree(c->d_sinfo);
        free(c->itp_lp);
        if (c->ktimer) {  // per-kernel timing events (gpmi_set_option "kernel_timing")
            KTimer *k = (KTimer *)c->ktimer;
            for (int cat = 0; cat < 3; ++cat) {
                for (hipEvent_t e : k->a[cat])
                    if (e) (void)hipEventDestroy(e);
                for (hipEvent_t e : k->b[cat])
                    if (e) (void)hipEventDestroy(e);
            }
            delete k;
        }
        hipEvent_t evs[] = {c->ev[0], c->ev[1], c->ev[2], c->ev[3], c->evM, c->evFork, c->evJoin, c->evP, c->evU};
        for (hipEvent_t e : evs)
            if (e) (void)hipEventDestroy(e);
        if (c->pstream) {
            (void)hipStreamSynchronize(c->pstream);
            (void)hipStreamDestroy(c->pstream);
        }
        for (int q = 0; q < c->nq; ++q)
            if (c->qstream[q]) (void)hipStreamDestroy(c->qstream[q]);
        if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    }
    free(c);
    return 0;
}

extern "C" int gpmi_set_stream(gpmi_ctx *c, void *hip_stream)
{
    ENTER(c);
    // handle 0 is HIP's legacy null stream (torch's default stream has cuda_stream == 0): work is
    // then ordered with everything else the caller enqueues there
    c->stream = (hipStream_t)hip_stream;
    return 0;
}

extern "C" int gpmi_reset_stream(gpmi_ctx *c)
{
    ENTER(c);
    c->stream = c->own_stream;
    return 0;
}

extern "C" int gpmi_sync(gpmi_ctx *c)
{
    ENTER(c);
    HIPCHK(hipStreamSynchronize(c->stream));
    return 0;
}

extern "C" int gpmi_set_option(gpmi_ctx *c, const char *name, int value)
{
    ENTER(c);
    if (!name) return gpmi_fail(GPMI_EARG, "option name is NULL");
    if (!strcmp(name, "nb_outer")) {
        if (value != 0 && (value < GPMI_NB || value % GPMI_NB))
            return gpmi_fail(GPMI_EARG, "nb_outer must be 0 (auto) or a multiple of %d", GPMI_NB);
        c->nb_outer = value;
        return 0;
    }
    if (!strcmp(name, "stagger")) {
        c->tune.stagger = value;
        return 0;
    }
    if (!strcmp(name, "se_nt")) {
        c->tune.se_nt = value != 0;
        return 0;
    }
    if (!strcmp(name, "ksplit")) {  // 0: off; 1: on; R > 1: on, split the tail round when it holds <= R tiles
        c->tune.ksplit = value != 0;
        if (value > 1) c->tune.ksplit_max = value;
        return 0;
    }
    if (!strcmp(name, "fuse_diag")) {
        c->tune.fuse_diag = value;
        return 0;
    }
    if (!strcmp(name, "block_recursive")) {
        c->tune.block_recursive = value != 0;
        return 0;
    }
    if (!strcmp(name, "nb_adapt")) {
        c->tune.nb_adapt = value != 0;
        return 0;
    }
    if (!strcmp(name, "nb_thr1024") || !strcmp(name, "nb_thr512") || !strcmp(name, "nb_thr256")) {
        if (value < 0) return gpmi_fail(GPMI_EARG, "%s must be >= 0", name);
        c->tune.nb_thr[name[6] == '1' ? 0 : (name[6] == '5' ? 1 : 2)] = value;
        return 0;
    }
    if (!strcmp(name, "syrk_order")) {   // tile walk of multi-round trailing updates: 0 row-major, 2 XCD-partitioned bands
#ifdef GPMI_PROBES
        if (value == 1) {                // (probe build) the padded 8 x 8 super-tile grid of round 1
            c->tune.syrk_order = 1;
            return 0;
        }
#endif
        if (value != 0 && value != 2) return gpmi_fail(GPMI_EARG, "syrk_order must be 0 (row-major) or 2 (XCD bands)");
        c->tune.syrk_order = value;
        return 0;
    }
    if (!strcmp(name, "small_ng1") || !strcmp(name, "small_ng")) {  // one-workgroup value + gradient: one evaluation / several
        if (value < 0 || value > 256) return gpmi_fail(GPMI_EARG, "%s must be 0 .. 256", name);
        (name[8] == '1' ? c->tune.small_ng1 : c->tune.small_ng) = value;
        return 0;
    }
    if (!strcmp(name, "grad_aug_n") || !strcmp(name, "grad_aug_ng")) {  // value + gradient through ONE augmented partial factorisation
        if (value < 0) return gpmi_fail(GPMI_EARG, "%s must be >= 0", name);   // up to this n: one evaluation / several at once (0: off)
        (name[10] == 'g' ? c->tune.grad_aug_ng : c->tune.grad_aug_n) = value;
        return 0;
    }
    if (!strcmp(name, "small_gc")) {  // gp_condition by one workgroup up to this many rows n + m + 1 (0: off)
        if (value < 0 || value > 1024) return gpmi_fail(GPMI_EARG, "small_gc must be 0 .. 1024");
        c->tune.small_gc = value;
        return 0;
    }
    if (!strcmp(name, "small_sd")) {  // sample_derivs[_batch]: one workgroup per draw up to this many rows n + m + 1 (0: off)
        if (value < 0 || value > 1024) return gpmi_fail(GPMI_EARG, "small_sd must be 0 .. 1024");
        c->tune.small_sd = value;
        return 0;
    }
    if (!strcmp(name, "small_sdb")) {  // ... for at least small_sdb ((n + m + 1) / 400)^2 draws
        if (value < 0) return gpmi_fail(GPMI_EARG, "small_sdb must be >= 0");
        c->tune.small_sdb = value;
        return 0;
    }
    if (!strcmp(name, "small_n2")) {  // ... up to this n for grids of at least small_g2 (n / 1024)^2 + 2 points (0: off)
        if (value < 0 || value > GPMI_SMALL_NMAX) return gpmi_fail(GPMI_EARG, "small_n2 must be 0 .. %d", GPMI_SMALL_NMAX);
        c->tune.small_n2 = value;
        return 0;
    }
    if (!strcmp(name, "small_g2")) {
        if (value < 0) return gpmi_fail(GPMI_EARG, "small_g2 must be >= 0");
        c->tune.small_g2 = value;
        return 0;
    }
    if (!strcmp(name, "small_n")) {  // one-workgroup marginal likelihood up to this n (0: off, <= 256)
        if (value < 0 || value > 256) return gpmi_fail(GPMI_EARG, "small_n must be 0 .. 256");
        c->tune.small_n = value;
        return 0;
    }
    if (!strcmp(name, "small_n1")) {  // ... for ONE evaluation (grids of fewer than 6 points)
        if (value < 0 || value > 256) return gpmi_fail(GPMI_EARG, "small_n1 must be 0 .. 256");
        c->tune.small_n1 = value;
        return 0;
    }
    if (!strcmp(name, "small_m")) {  // one-workgroup partial factorisation up to this many rows (0: off)
        if (value < 0 || value > 1024) return gpmi_fail(GPMI_EARG, "small_m must be 0 .. 1024");
        c->tune.small_m = value;
        return 0;
    }
#ifdef GPMI_PROBES
    // switches of measured-and-rejected variants, kept as A/B material in the probe build only
    if (!strcmp(name, "diag_waves")) {   // 5: the 5-wave / 3-barrier diagonal-block kernel of round 1
        if (value != 4 && value != 5) return gpmi_fail(GPMI_EARG, "diag_waves must be 4 or 5");
        c->tune.diag_waves = value;
        return 0;
    }
    if (!strcmp(name, "lookahead")) {    // 1: two-stream look-ahead over outer blocks inside one factorisation
        c->lookahead = value > 0;
        return 0;
    }
    if (!strcmp(name, "lane_lookahead")) {
        c->lane_lookahead = value != 0;
        return 0;
    }
    if (!strcmp(name, "rect_auto")) {
        c->tune.rect_auto = value;
        return 0;
    }
    if (!strcmp(name, "gemm_variant")) {
        c->tune.gemm_variant = value;
        return 0;
    }
    if (!strcmp(name, "debug_topology")) {  // prints which of the context's streams dispatch concurrently
        hipStream_t st[4] = {c->stream, c->own_stream, c->pstream, c->nq > 1 ? c->qstream[1] : nullptr};
        const char *nm[4] = {"stream", "own", "panel", "q1"};
        for (int a = 0; a < 4; ++a) {
            fprintf(stderr, "%8s:", nm[a]);
            for (int b = 0; b < 4; ++b) {
                bool conc = false;
                if (a != b && st[a] && st[b] && st[a] != st[b]) {
                    int rc = streams_concurrent(c, st[a], st[b], &conc);
                    if (rc) return rc;
                }
                fprintf(stderr, " %c", (a == b || st[a] == st[b]) ? '-' : (conc ? 'c' : 'S'));
            }
            fprintf(stderr, "\n");
        }
        return 0;
    }
#endif  // GPMI_PROBES
    if (!strcmp(name, "calibrate")) {
        c->calibrate = value != 0;
        return 0;
    }
    if (!strcmp(name, "grid_lanes")) {
        if (value < 0 || value > 8) return gpmi_fail(GPMI_EARG, "grid_lanes must be 0 (auto) .. 8");
        c->grid_lanes = value;
        return 0;
    }
    if (!strcmp(name, "timing")) {
        c->timing = value != 0;
        return 0;
    }
    if (!strcmp(name, "kernel_timing")) {
        if (!c->ktimer) c->ktimer = new KTimer();
        c->ktiming = value != 0;
        return 0;
    }
    return gpmi_fail(GPMI_EARG, "unknown option '%s'", name);
}

// workspace for an (rows x cols) lower-stored matrix; ld padded so that columns do not
// alias HBM channels and 16-column slack exists past the end for tile over-reads
static int reserve_ws(gpmi_ctx *c, int rows, int cols)
{
    int ld = ((rows + 15) / 16) * 16 + 16;
#ifdef GPMI_PROBES
    if (const char *e = getenv("GPMI_LD")) {  // experiment: force the leading dimension (>= the natural one)
        const int v = atoi(e);
        if (v >= ld) ld = v;
    }
    if (const char *e = getenv("GPMI_LD_PAD")) ld += atoi(e);
#endif
    const size_t need = ((size_t)ld * (size_t)(cols + 1) + 4096) * sizeof(double);
    if (need > c->W_bytes) {
        HIPCHK(hipStreamSynchronize(c->stream));
        if (c->W) HIPCHK(hipFree(c->W));
        c->W = nullptr;
        c->W_bytes = 0;
        if (hipMalloc((void **)&c->W, need) != hipSuccess)
            return gpmi_fail(GPMI_ENOMEM, "cannot allocate %zu bytes of workspace", need);
        c->W_bytes = need;
    }
    c->ld = ld;
    c->ncols = cols;
    return 0;
}

extern "C" int gpmi_reserve(gpmi_ctx *c, int n_max)
{
    ENTER(c);
    if (n_max <= 0) return gpmi_fail(GPMI_EARG, "n_max must be positive");
    int rc = reserve_ws(c, n_max + 1, n_max + 1);
    if (rc) return rc;
    // also the grid lanes (contexts, streams, workspaces), so that no allocation happens later
    // inside a grid call: up to 4 lanes are used in auto mode
    const int lanes = c->grid_lanes > 0 ? c->grid_lanes : 4;
    for (int l = 1; l < lanes && l <= 7; ++l) {
        if (!c->lane[l - 1] && (rc = gpmi_create(&c->lane[l - 1], c->device))) return rc;
        if ((rc = reserve_ws(c->lane[l - 1], n_max + 1, n_max + 1))) return rc;
    }
    if (lanes > 1 && c->calibrate && (rc = calibrate_streams(c, lanes))) return rc;
    return 0;
}

static int stage_buf(gpmi_ctx *c, int slot, size_t bytes, double **out)
{
    bytes += 4096 * sizeof(double);  // slack for tile over-reads
    if (bytes > c->stage_bytes[slot]) {
        HIPCHK(hipStreamSynchronize(c->stream));
        if (c->stage[slot]) HIPCHK(hipFree(c->stage[slot]));
        c->stage[slot] = nullptr;
        c->stage_bytes[slot] = 0;
        if (hipMalloc((void **)&c->stage[slot], bytes) != hipSuccess)
            return gpmi_fail(GPMI_ENOMEM, "cannot allocate %zu bytes of staging", bytes);
        c->stage_bytes[slot] = bytes;
    }
    *out = c->stage[slot];
    return 0;
}

static int scratch_buf(gpmi_ctx *c, size_t bytes, double **out)
{
    if (bytes > c->scratch_bytes) {
        HIPCHK(hipStreamSynchronize(c->stream));
        if (c->scratch) HIPCHK(hipFree(c->scratch));
        c->scratch = nullptr;
        c->scratch_bytes = 0;
        if (hipMalloc((void **)&c->scratch, bytes) != hipSuccess)
            return gpmi_fail(GPMI_ENOMEM, "cannot allocate %zu bytes of scratch", bytes);
        c->scratch_bytes = bytes;
    }
    *out = c->scratch;
    return 0;
}

static int fill_params(SeParams *p, int D, double alpha, const double *ell, int n_ell)
{
    if (D < 1 || D > GPMI_MAXD_BIG) return gpmi_fail(GPMI_EARG, "D = %d unsupported (1..%d)", D, GPMI_MAXD_BIG);
    if (!ell || (n_ell != 1 && n_ell != D)) return gpmi_fail(GPMI_EARG, "length-scale vector must have length 1 or D");
    p->a2 = alpha * alpha;
    p->D = D;
    for (int d = 0; d < GPMI_MAXD_BIG; ++d) p->inv_ell[d] = 0.0;
    for (int d = 0; d < D; ++d) {
        const double l = ell[n_ell == 1 ? 0 : d];
        if (!(l > 0.0)) return gpmi_fail(GPMI_EARG, "length-scale must be positive");
        p->inv_ell[d] = 1.0 / l;
    }
    return 0;
}

// copy a host column-major block (rows x cols, ld) to a packed device buffer (ld = rows)
static int h2d_matrix(gpmi_ctx *c, const double *h, int rows, int cols, int ldh, double *d)
{
    if (rows <= 0 || cols <= 0) return 0;
    HIPCHK(hipMemcpy2DAsync(d, (size_t)rows * sizeof(double), h, (size_t)ldh * sizeof(double),
                            (size_t)rows * sizeof(double), cols, hipMemcpyHostToDevice, c->stream));
    return 0;
}
static int d2h_matrix(gpmi_ctx *c, const double *d, size_t ldd, int rows, int cols, double *h, int ldh)
{
    if (rows <= 0 || cols <= 0) return 0;
    HIPCHK(hipMemcpy2DAsync(h, (size_t)ldh * sizeof(double), d, ldd * sizeof(double),
                            (size_t)rows * sizeof(double), cols, hipMemcpyDeviceToHost, c->stream));
    return 0;
}

// ---- covariance builders ------------------------------------------------------
extern "C" int gpmi_se_cov_dev(gpmi_ctx *c, const double *dX, int n, int ldx, const double *dY, int m,
                               int ldy, int D, double alpha, const double *ell, int n_ell,
                               double diag_add, int flags, double *dK, int ldk)
{
    ENTER(c);
    if (n < 0 || m < 0) return gpmi_fail(GPMI_EARG, "negative size");
    if (!dY) m = n;
    if (n == 0 || m == 0) return 0;
    if (!dX || !dK || ldx < n || ldk < n || (dY && ldy < m)) return gpmi_fail(GPMI_EARG, "bad pointer or leading dimension");
    if ((flags & GPMI_LOWER) && dY) return gpmi_fail(GPMI_EARG, "GPMI_LOWER needs Y == NULL (symmetric case)");
    SeParams p;
    int rc = fill_params(&p, D, alpha, ell, n_ell);
    if (rc) return rc;
    launch_se_cov(c, c->stream, dX, n, ldx, dY, m, ldy, p, diag_add, (flags & GPMI_LOWER) ? 1 : 0, dK, (size_t)ldk);
    HIPCHK(hipGetLastError());
    return 0;
}

extern "C" int gpmi_se_cov(gpmi_ctx *c, const double *X, int n, int ldx, const double *Y, int m, int ldy,
                           int D, double alpha, const double *ell, int n_ell, double diag_add, int flags,
                           double *K, int ldk)
{
    ENTER(c);
    if (n < 0 || m < 0) return gpmi_fail(GPMI_EARG, "negative size");
    if (!Y) m = n;
    if (n == 0 || m == 0) return 0;
    if (!X || !K || ldx < n || ldk < n || (Y && ldy < m) || D < 1) return gpmi_fail(GPMI_EARG, "bad pointer or leading dimension");
    double *dX, *dY = nullptr, *dK;
    int rc;
    if ((rc = stage_buf(c, 0, (size_t)n * D * sizeof(double), &dX))) return rc;
    if (Y && (rc = stage_buf(c, 1, (size_t)m * D * sizeof(double), &dY))) return rc;
    const int ldd = (n + 1) & ~1;
    if ((rc = stage_buf(c, 2, (size_t)ldd * m * sizeof(double), &dK))) return rc;
    if ((rc = h2d_matrix(c, X, n, D, ldx, dX))) return rc;
    if (Y && (rc = h2d_matrix(c, Y, m, D, ldy, dY))) return rc;
    if (flags & GPMI_LOWER) HIPCHK(hipMemsetAsync(dK, 0, (size_t)ldd * m * sizeof(double), c->stream));
    if ((rc = gpmi_se_cov_dev(c, dX, n, n, dY, m, m, D, alpha, ell, n_ell, diag_add, flags, dK, ldd))) return rc;
    if ((rc = d2h_matrix(c, dK, ldd, n, m, K, ldk))) return rc;
    HIPCHK(hipStreamSynchronize(c->stream));
    return 0;
}

extern "C" int gpmi_deriv_cov_dev(gpmi_ctx *c, int kind, const double *dx, int n, const double *dy, int m,
                                  double alpha, double l, int flags, double *dK, int ldk)
{
    ENTER(c);
    if (kind < 0 || kind > GPMI_TT) return gpmi_fail(GPMI_EARG, "bad kernel kind %d", kind);
    if (n < 0 || m < 0) return gpmi_fail(GPMI_EARG, "negative size");
    if (n == 0 || m == 0) return 0;
    if (!dx || !dy || !dK || ldk < n || !(l > 0.0)) return gpmi_fail(GPMI_EARG, "bad argument");
    launch_deriv_cov(c->stream, kind, dx, n, dy, m, alpha * alpha, l, (flags & GPMI_COMPAT_RR) ? 1 : 0,
                     (flags & GPMI_LOWER) ? 1 : 0, dK, (size_t)ldk);
    HIPCHK(hipGetLastError());
    return 0;
}

extern "C" int gpmi_deriv_cov(gpmi_ctx *c, int kind, const double *x, int n, const double *y, int m,
                              double alpha, double l, int flags, double *K, int ldk)
{
    ENTER(c);
    if (n < 0 || m < 0) return gpmi_fail(GPMI_EARG, "negative size");
    if (n == 0 || m == 0) return 0;
    if (!x || !y || !K || ldk < n) return gpmi_fail(GPMI_EARG, "bad argument");
    double *dx, *dy, *dK;
    int rc;
    if ((rc = stage_buf(c, 0, (size_t)n * sizeof(double), &dx))) return rc;
    if ((rc = stage_buf(c, 1, (size_t)m * sizeof(double), &dy))) return rc;
    const int ldd = (n + 1) & ~1;
    if ((rc = stage_buf(c, 2, (size_t)ldd * m * sizeof(double), &dK))) return rc;
    HIPCHK(hipMemcpyAsync(dx, x, (size_t)n * sizeof(double), hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(dy, y, (size_t)m * sizeof(double), hipMemcpyHostToDevice, c->stream));
    if (flags & GPMI_LOWER) HIPCHK(hipMemsetAsync(dK, 0, (size_t)ldd * m * sizeof(double), c->stream));
    if ((rc = gpmi_deriv_cov_dev(c, kind, dx, n, dy, m, alpha, l, flags, dK, ldd))) return rc;
    if ((rc = d2h_matrix(c, dK, ldd, n, m, K, ldk))) return rc;
    HIPCHK(hipStreamSynchronize(c->stream));
    return 0;
}

extern "C" int gpmi_deriv_elem(gpmi_ctx *c, int kind, const double *tj, const double *tk, size_t len,
                               double l, double *out)
{
    ENTER(c);
    if (kind < 0 || kind > GPMI_TT) return gpmi_fail(GPMI_EARG, "bad kernel kind %d", kind);
    if (len == 0) return 0;
    if (!tj || !tk || !out || !(l > 0.0)) return gpmi_fail(GPMI_EARG, "bad argument");
    double *a, *b, *o;
    int rc;
    if ((rc = stage_buf(c, 0, len * sizeof(double), &a))) return rc;
    if ((rc = stage_buf(c, 1, len * sizeof(double), &b))) return rc;
    if ((rc = stage_buf(c, 2, len * sizeof(double), &o))) return rc;
    HIPCHK(hipMemcpyAsync(a, tj, len * sizeof(double), hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(b, tk, len * sizeof(double), hipMemcpyHostToDevice, c->stream));
    launch_deriv_elem(c->stream, kind, a, b, len, l, o);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(out, o, len * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return 0;
}

extern "C" int gpmi_joint_cov(gpmi_ctx *c, const double *t, int n, double alpha, double l, double sigma,
                              double jitter, int flags, double *K, int ldk)
{
    ENTER(c);
    if (n < 0) return gpmi_fail(GPMI_EARG, "negative size");
    if (n == 0) return 0;
    if (!t || !K || ldk < 2 * n || !(l > 0.0)) return gpmi_fail(GPMI_EARG, "bad argument");
    double *dt, *dK;
    int rc;
    const int n2 = 2 * n;
    if ((rc = stage_buf(c, 0, (size_t)n * sizeof(double), &dt))) return rc;
    if ((rc = stage_buf(c, 2, (size_t)n2 * n2 * sizeof(double), &dK))) return rc;
    HIPCHK(hipMemcpyAsync(dt, t, (size_t)n * sizeof(double), hipMemcpyHostToDevice, c->stream));
    if (flags & GPMI_LOWER) HIPCHK(hipMemsetAsync(dK, 0, (size_t)n2 * n2 * sizeof(double), c->stream));
    launch_joint_cov(c->stream, dt, n, alpha * alpha, l, sigma * sigma, jitter, (flags & GPMI_COMPAT_RR) ? 1 : 0,
                     (flags & GPMI_LOWER) ? 1 : 0, dK, (size_t)n2);
    HIPCHK(hipGetLastError());
    if ((rc = d2h_matrix(c, dK, n2, n2, n2, K, ldk))) return rc;
    HIPCHK(hipStreamSynchronize(c->stream));
    return 0;
}

// ---- factorisation ------------------------------------------------------------
static void tic(gpmi_ctx *c, int i)
{
    if (c->timing) (void)hipEventRecord(c->ev[i], c->stream);
}

extern "C" int gpmi_potrf_dev(gpmi_ctx *c, double *dA, int n, int lda, int *d_info)
{
    ENTER(c);
    if (n < 0) return gpmi_fail(GPMI_EARG, "negative size");
    if (!d_info) return gpmi_fail(GPMI_EARG, "d_info is NULL");
    HIPCHK(hipMemsetAsync(d_info, 0, sizeof(int), c->stream));
    if (n == 0) return 0;
    if (!dA || lda < n) return gpmi_fail(GPMI_EARG, "bad argument");
    int rc;
    if ((rc = reserve_ws(c, n, n))) return rc;
    launch_copy_matrix(c->stream, dA, (size_t)lda, c->W, (size_t)c->ld, n, n, 1);
    if ((rc = launch_potrf_partial(c, c->W, (size_t)c->ld, n, n, n, d_info, nullptr))) return rc;
    launch_copy_matrix(c->stream, c->W, (size_t)c->ld, dA, (size_t)lda, n, n, 1);
    HIPCHK(hipGetLastError());
    return 0;
}

extern "C" int gpmi_potrf(gpmi_ctx *c, double *A, int n, int lda)
{
    ENTER(c);
    if (n < 0) return gpmi_fail(GPMI_EARG, "negative size");
    if (n == 0) return 0;
    if (!A || lda < n) return gpmi_fail(GPMI_EARG, "bad argument");
    double *dA;
    int rc;
    if ((rc = stage_buf(c, 2, (size_t)n * n * sizeof(double), &dA))) return rc;
    if ((rc = h2d_matrix(c, A, n, n, lda, dA))) return rc;
    if ((rc = gpmi_potrf_dev(c, dA, n, n, c->d_info))) return rc;
    int info = 0;
    HIPCHK(hipMemcpyAsync(&info, c->d_info, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    if ((rc = d2h_matrix(c, dA, n, n, n, A, lda))) return rc;
    HIPCHK(hipStreamSynchronize(c->stream));
    return info;
}

extern "C" int gpmi_trmv_lower(gpmi_ctx *c, const double *L, int n, int ldl, const double *z, double *f)
{
    ENTER(c);
    if (n < 0) return gpmi_fail(GPMI_EARG, "negative size");
    if (n == 0) return 0;
    if (!L || !z || !f || ldl < n) return gpmi_fail(GPMI_EARG, "bad argument");
    double *dL, *dz, *df;
    int rc;
    if ((rc = stage_buf(c, 2, (size_t)n * n * sizeof(double), &dL))) return rc;
    if ((rc = stage_buf(c, 0, (size_t)n * sizeof(double), &dz))) return rc;
    if ((rc = stage_buf(c, 1, (size_t)n * (1 + trmv_lower_chunks(n)) * sizeof(double), &df))) return rc;
    if ((rc = h2d_matrix(c, L, n, n, ldl, dL))) return rc;
    HIPCHK(hipMemcpyAsync(dz, z, (size_t)n * sizeof(double), hipMemcpyHostToDevice, c->stream));
    launch_trmv_lower(c->stream, dL, (size_t)n, n, dz, df, df + n);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(f, df, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return 0;
}

// z = L^-1 b for a given lower factor (mdivide_left_tri_low): the one-launch wavefront solve (k_trsv_wave);
// the block factors and the inverses of the 128 x 128 diagonal blocks are packed from L itself.
extern "C" int gpmi_trsv_lower(gpmi_ctx *c, const double *L, int n, int ldl, const double *b, double *z)
{
    ENTER(c);
    if (n < 0) return gpmi_fail(GPMI_EARG, "negative size");
    if (n == 0) return 0;
    if (!L || !b || !z || ldl < n) return gpmi_fail(GPMI_EARG, "bad argument");
    int rc;
    const int ldd = ((n + 15) / 16) * 16 + 16;
    const int npan = (n + GPMI_NB - 1) / GPMI_NB;
    double *dL, *db, *dx, *Fall;
    const size_t dinv = (size_t)npan * GPMI_NB * GPMI_NB;  // inverse diagonal blocks + the same of scratch, behind the factors
    if ((rc = stage_buf(c, 2, (size_t)ldd * (n + 1) * sizeof(double), &dL))) return rc;
    if ((rc = stage_buf(c, 0, (size_t)n * sizeof(double), &db))) return rc;
    if ((rc = stage_buf(c, 1, (size_t)n * sizeof(double), &dx))) return rc;
    if ((rc = scratch_buf(c, ((size_t)npan * GPMI_FPACK + 2 * dinv) * sizeof(double), &Fall))) return rc;
    double *Dinv = Fall + (size_t)npan * GPMI_FPACK;
    hipStream_t s = c->stream;
    HIPCHK(hipMemcpy2DAsync(dL, (size_t)ldd * sizeof(double), L, (size_t)ldl * sizeof(double),
                            (size_t)n * sizeof(double), n, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(db, b, (size_t)n * sizeof(double), hipMemcpyHostToDevice, s));
    launch_pack_factors(s, dL, (size_t)ldd, n, Fall);
    launch_diag_inverses(s, Fall, n, Dinv, Dinv + dinv);
    if ((rc = launch_trsv_lower(s, dL, (size_t)ldd, n, db, dx, Dinv, c->d_ctr + 32))) return rc;  // one launch, the factor read once
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(z, dx, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    return 0;
}

// ---- lanes: concurrent independent factorisations inside one GPU ---------------------------
// Lane 0 is the context itself, lanes 1.. are internal contexts with their own workspaces.  Used
// by the hyper-parameter grid and by the interpolation-table build (independent length-scales).
static int lanes_prepare(gpmi_ctx *c, int lanes)
{
    for (int l = 1; l < lanes; ++l) {
        if (!c->lane[l - 1]) {
            int rc = gpmi_create(&c->lane[l - 1], c->device);
            if (rc) return rc;
        }
        gpmi_ctx *lc = c->lane[l - 1];
        lc->tune = c->tune;
        lc->nb_outer = c->nb_outer;
        lc->lookahead = c->lane_lookahead;  // default 0: concurrent lanes already fill the panel phases
        lc->calibrate = 0;                  // a lane never probes for streams of its own
    }
    if (lanes > 1 && c->calibrate) {
        int rc = calibrate_streams(c, lanes);
        if (rc) return rc;
    }
    return 0;
}

// lanes fork from / join into the caller's stream; with calibration every lane (lane 0 is this
// context itself) runs on a stream of its own hardware pipe
static void lanes_fork(gpmi_ctx *c, int lanes, hipStream_t caller)
{
    if (lanes <= 1) return;
    c->lookahead = c->lane_lookahead;
    const bool useq = c->calibrate && c->nq > 1;
    (void)hipEventRecord(c->evFork, caller);
    for (int l = 0; l < lanes; ++l) {
        gpmi_ctx *lc = l ? c->lane[l - 1] : c;
        lc->stream = (useq && l < c->nq) ? c->qstream[l] : (l ? lc->own_stream : caller);
        lc->lanes_active = 1;
        if (lc->stream != caller) (void)hipStreamWaitEvent(lc->stream, c->evFork, 0);
    }
}

static void lanes_join(gpmi_ctx *c, int lanes, hipStream_t caller, int la_saved)
{
    if (lanes <= 1) return;
    c->lookahead = la_saved;
    for (int l = 0; l < lanes; ++l) {
        gpmi_ctx *lc = l ? c->lane[l - 1] : c;
        if (lc->stream != caller) {
            (void)hipEventRecord(lc->evJoin, lc->stream);
            (void)hipStreamWaitEvent(caller, lc->evJoin, 0);
        }
        lc->stream = l ? lc->own_stream : caller;
        lc->lanes_active = 0;
    }
}

// ---- latent exact GP: f = chol(K) z --------------------------------------------------------
static int pin_reserve(gpmi_ctx *c, size_t need);
static int pin_wait(gpmi_ctx *c, const int *flag, int seq);
static int reserve_ws_small(gpmi_ctx *c, int n, int count);
static int upload_xy(gpmi_ctx *c, const double *X, int n, int ldx, int D, const double *y, double **dX, double **dy);
extern "C" int gpmi_exact_gp_f(gpmi_ctx *c, const double *X, int n, int ldx, int D, double alpha, const double *ell, int n_ell,
                               double jitter, const double *z, double *f)
{
    ENTER(c);
    if (n <= 0 || !X || !z || !f || ldx < n || D < 1) return gpmi_fail(GPMI_EARG, "bad argument");
    SeParams p;
    int rc;
    if ((rc = fill_params(&p, D, alpha, ell, n_ell))) return rc;
    hipStream_t s = c->stream;
    if (n <= 256 && D <= GPMI_MAXD) {
        // one launch of one workgroup; X, z in and f, info out through the pinned, device-mapped buffer: [info | f | X | z]
        const size_t need = (8 + (size_t)n * (D + 2)) * sizeof(double);
        if ((rc = pin_reserve(c, need))) return rc;
        double *hf = c->h_pin + 8, *hX = hf + n, *hz = hX + (size_t)n * D, *stage;
        for (int d = 0; d < D; ++d) memcpy(hX + (size_t)d * n, X + (size_t)d * ldx, (size_t)n * sizeof(double));
        memcpy(hz, z, (size_t)n * sizeof(double));
        if ((rc = scratch_buf(c, (size_t)n * (D + 1) * sizeof(double), &stage))) return rc;
        if ((rc = reserve_ws_small(c, n, 1))) return rc;
        double *pd = c->h_pin_dev;
        launch_exact_gp_small(s, pd + 8 + n, n, n, pd + 8 + n + (size_t)n * D, p, jitter, c->W, pd + 8, (int *)pd, c->d_info, stage,
                              (int *)(pd + 7), ++c->pin_seq);
        HIPCHK(hipGetLastError());
        if ((rc = pin_wait(c, (const int *)(c->h_pin + 7), c->pin_seq))) return rc;
        memcpy(f, hf, (size_t)n * sizeof(double));
        return *(const int *)c->h_pin;
    }
    double *dX, *dz;
    if ((rc = upload_xy(c, X, n, ldx, D, z, &dX, &dz))) return rc;
    if ((rc = reserve_ws(c, n, n))) return rc;
    const size_t ld = (size_t)c->ld;
    double *part;
    if ((rc = stage_buf(c, 3, ((size_t)trmv_lower_chunks(n) + 1) * n * sizeof(double), &part))) return rc;
    double *df = part + (size_t)trmv_lower_chunks(n) * n;
    HIPCHK(hipMemsetAsync(c->d_info, 0, sizeof(int), s));
    launch_se_cov(c, s, dX, n, n, nullptr, n, n, p, jitter, 1, c->W, ld);
    if ((rc = launch_potrf_partial(c, c->W, ld, n, n, n, c->d_info, nullptr))) return rc;
    launch_trmv_lower(s, c->W, ld, n, dz, df, part);
    int info = 0;
    HIPCHK(hipMemcpyAsync(f, df, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(&info, c->d_info, sizeof(int), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    if (info)
        for (int i = 0; i < n; ++i) f[i] = NAN;
    return info;
}

// ---- marginal likelihood -----------------------------------------------------
// n small enough for the one-workgroup evaluation (k_logml_small): the sizes the reference's own drivers run
// at (R/tests.R:5 N = 21, pendulum_fit*.R 79 .. 199, BASELINE c1 N = 256)
// pinned, device-mapped host buffer of the one-launch host-buffer calls (inputs in, results out, no copy call)
static int pin_reserve(gpmi_ctx *c, size_t need)
{
    if (need <= c->h_pin_bytes) return 0;
    HIPCHK(hipStreamSynchronize(c->stream));
    if (c->h_pin) HIPCHK(hipHostFree(c->h_pin));
    c->h_pin = c->h_pin_dev = nullptr;
    c->h_pin_bytes = 0;
    const size_t want = need > 65536 ? need : 65536;
    if (hipHostMalloc((void **)&c->h_pin, want, hipHostMallocMapped) != hipSuccess)
        return gpmi_fail(GPMI_ENOMEM, "cannot allocate %zu bytes of pinned host memory", want);
    HIPCHK(hipHostGetDevicePointer((void **)&c->h_pin_dev, c->h_pin, 0));
    c->h_pin_bytes = want;
    return 0;
}

// wait for the completion flag a one-launch kernel publishes in the pinned buffer (small_signal_done): polling the mapped
// word costs ~1-2 us after the kernel's last store, a stream synchronisation ~10; the stream is the fallback
static int pin_wait(gpmi_ctx *c, const int *flag, int seq)
{
    for (long it = 0; it < 20000000L; ++it) {
        if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) == seq) return 0;
        __builtin_ia32_pause();
    }
    HIPCHK(hipStreamSynchronize(c->stream));
    if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) == seq) return 0;
    return gpmi_fail(GPMI_EHIP, "the kernel of a one-launch call did not publish its completion flag");
}

static bool small_logml(const gpmi_ctx *c, int n, int D, int G = 1)
{
    // one workgroup against the multi-CU launch chain (tools/small_n_bench.py, one evaluation / 64 points, us per
    // evaluation): n = 21: 20 vs 36 / 0.6 vs 22; 128: 43 vs 43 / 1.0 vs 23; 199: 118 vs 89 / 2.2 vs 33; 256: 145 vs 79 / 2.6 vs 30
    int lim = (G >= 6 || c->tune.small_n1 > c->tune.small_n) ? c->tune.small_n : c->tune.small_n1;
    // larger grids at mid sizes: a workgroup needs n^3 time, but G of them run side by side on CUs of their own, the
    // four lanes of the blocked path one point after another at ~36 us per 128-column panel
    // (tools/mid_n_grid_bench.py: one workgroup takes 0.21 / 0.59 / 1.55 / 3.1 ms at n = 300 / 512 / 768 / 1024 whatever
    // the number of points up to 256, the lanes 54 / 60 / 85 / 109 us PER POINT: break-even at G = 5 / 12 / 24 / 38,
    // i.e. about small_g2 (n / 1024)^2 + 2 points with small_g2 = 40)
    if (lim > 0 && n > lim && n <= c->tune.small_n2) {
        const double r = (double)n / 1024.0;
        if ((double)G >= (double)c->tune.small_g2 * r * r + 2.0) lim = c->tune.small_n2;
    }
    return lim > 0 && n <= lim && n <= GPMI_SMALL_NMAX && D <= GPMI_MAXD;
}

// device buffers of the device-parameter small-N grid: parameters and one work int per point
static int reserve_small_par(gpmi_ctx *c, int G)
{
    if (G <= c->spar_pts) return 0;
    HIPCHK(hipStreamSynchronize(c->stream));
    if (c->d_spar) HIPCHK(hipFree(c->d_spar));
    if (c->d_sinfo) HIPCHK(hipFree(c->d_sinfo));
    c->d_spar = nullptr;
    c->d_sinfo = nullptr;
    c->spar_pts = 0;
    const int cap = G < 1024 ? 1024 : G;
    if (hipMalloc((void **)&c->d_spar, (size_t)cap * GPMI_SMALL_PAR * sizeof(double)) != hipSuccess ||
        hipMalloc((void **)&c->d_sinfo, (size_t)cap * sizeof(int)) != hipSuccess)
        return gpmi_fail(GPMI_ENOMEM, "cannot allocate the parameter buffers of a %d-point grid", G);
    c->spar_pts = cap;
    return 0;
}

// G points by one workgroup each: up to `per_args` of them per launch with the parameters as kernel arguments (lowest
// latency), larger grids and n > 256 through the device-parameter form (any number per launch, GPMI_SMALL_DEV_PTS
// workspace slices at a time)
static int small_grid(gpmi_ctx *c, const double *dX, int n, int ldx, int D, const double *dy, const double *alpha,
                      const double *ell, int n_ell, const double *sigma, int G, double jitter, double *d_out3, int *d_info)
{
    int rc;
    const int per_args = n_ell == 1 ? GPMI_SMALL_PTS : GPMI_SMALL_PTS_ARD;
    if (G <= per_args && n <= 256) {
        if ((rc = reserve_ws_small(c, n, G))) return rc;
        if (n_ell == 1) launch_logml_small_batch(c->stream, dX, n, ldx, D, dy, alpha, ell, sigma, G, jitter, c->W, d_out3, d_info, c->d_ctr + 64);
        else launch_logml_small_batch_ard(c->stream, dX, n, ldx, D, dy, alpha, ell, sigma, G, jitter, c->W, d_out3, d_info, c->d_ctr + 64);
        HIPCHK(hipGetLastError());
        return 0;
    }
    const int per = G < GPMI_SMALL_DEV_PTS ? G : GPMI_SMALL_DEV_PTS;
    if ((rc = reserve_ws_small(c, n, per))) return rc;
    if ((rc = reserve_small_par(c, per))) return rc;
    for (int g0 = 0; g0 < G; g0 += per) {
        const int gc = (G - g0 < per) ? G - g0 : per;
        launch_logml_small_batch_dev(c->stream, dX, n, ldx, D, dy, alpha + g0, ell + (size_t)g0 * (n_ell == 1 ? 1 : D), n_ell,
                                     sigma + g0, gc, jitter, c->d_spar, c->W, d_out3 + 3 * (size_t)g0, d_info + g0, c->d_sinfo);
    }
    HIPCHK(hipGetLastError());
    return 0;
}

// workspace for `count` slices of the small-N kernel
static int reserve_ws_small(gpmi_ctx *c, int n, int count)
{
    size_t ld, stride;
    small_ws_layout(n, &ld, &stride);
    const size_t need = (stride * (size_t)count + 4096) * sizeof(double);
    if (need > c->W_bytes) {
        HIPCHK(hipStreamSynchronize(c->stream));
        if (c->W) HIPCHK(hipFree(c->W));
        c->W = nullptr;
        c->W_bytes = 0;
        if (hipMalloc((void **)&c->W, need) != hipSuccess)
            return gpmi_fail(GPMI_ENOMEM, "cannot allocate %zu bytes of workspace", need);
        c->W_bytes = need;
    }
    c->ld = (int)ld;
    c->ncols = n;
    return 0;
}

static int logml_core(gpmi_ctx *c, const double *dX, int n, int ldx, const double *dy, const SeParams &p,
                      double diag_add, double *d_out3, int *d_info)
{
    int rc;
    const int M = n + 1;
    if (small_logml(c, n, p.D)) {  // build, factorisation, solve and log-det in ONE launch of one workgroup
        if ((rc = reserve_ws_small(c, n, 1))) return rc;
        tic(c, 0);
        tic(c, 1);
        launch_logml_small(c->stream, dX, n, ldx, dy, p, diag_add, c->W, (size_t)c->ld, d_out3, d_info, c->d_info, nullptr);
        tic(c, 2);
        tic(c, 3);
        HIPCHK(hipGetLastError());
        return 0;
    }
    if ((rc = reserve_ws(c, M, n))) return rc;
    const size_t ld = (size_t)c->ld;
    tic(c, 0);
    HIPCHK(hipMemsetAsync(c->d_info, 0, sizeof(int), c->stream));
    kt_begin(c, 0);
    launch_se_cov(c, c->stream, dX, n, ldx, nullptr, n, ldx, p, diag_add, 1, c->W, ld);
    kt_end(c, 0, 4.0 * (double)n * ((double)n + 1.0));  // lower triangle incl. diagonal, 8 B each
    launch_set_row(c->stream, c->W, ld, n, dy, n, n);
    tic(c, 1);
    if ((rc = launch_potrf_partial(c, c->W, ld, M, n, n, c->d_info, nullptr))) return rc;
    tic(c, 2);
    launch_logml_finalize(c->stream, c->W, ld, n, n, c->d_info, d_out3, d_info, c->d_fin);
    tic(c, 3);
    HIPCHK(hipGetLastError());
    return 0;
}

extern "C" int gpmi_logml_dev(gpmi_ctx *c, const double *dX, int n, int ldx, int D, const double *dy,
                              double alpha, const double *ell, int n_ell, double sigma, double jitter,
                              double *d_out3, int *d_info)
{
    ENTER(c);
    if (n <= 0 || !dX || !dy || !d_out3 || ldx < n) return gpmi_fail(GPMI_EARG, "bad argument");
    SeParams p;
    int rc = fill_params(&p, D, alpha, ell, n_ell);
    if (rc) return rc;
    return logml_core(c, dX, n, ldx, dy, p, sigma * sigma + jitter, d_out3, d_info);
}

extern "C" int gpmi_logml_grid_dev(gpmi_ctx *c, const double *dX, int n, int ldx, int D, const double *dy,
                                   const double *alpha, const double *rho, const double *sigma, int G,
                                   double jitter, double *d_out3, int *d_info)
{
    ENTER(c);
    if (G < 0) return gpmi_fail(GPMI_EARG, "negative grid size");
    if (G == 0) return 0;
    if (n <= 0 || !dX || !dy || !alpha || !rho || !sigma || !d_out3 || !d_info || ldx < n)
        return gpmi_fail(GPMI_EARG, "bad argument");
    if (small_logml(c, n, D, G)) {
        // small n: every point is ONE workgroup; up to GPMI_SMALL_PTS points per launch (their hyper-parameters
        // travel as kernel arguments), no lanes, no per-point launch chain
        for (int g = 0; g < G; ++g)
            if (!(rho[g] > 0.0)) return gpmi_fail(GPMI_EARG, "length-scale must be positive");
        return small_grid(c, dX, n, ldx, D, dy, alpha, rho, 1, sigma, G, jitter, d_out3, d_info);
    }
    // Independent points fan out over `lanes` internal contexts (own workspaces and streams):
    // while one point is in its latency-bound panel phase or in the tail of a trailing update,
    // another point's bulk update fills the chip.  Lanes fork from / join into the caller's stream.
    // auto: 4 lanes (one per calibrated dispatch stream), or 3 when that splits the grid evenly and 4 does not; a fifth
    // lane has no pipe of its own: G = 10 at N = 4096 / 8192 took 0.85 / 3.89 ms per point on 5 lanes, 0.71 / 3.58 on 4
    int lanes = c->grid_lanes > 0 ? c->grid_lanes : (G % 4 == 0 ? 4 : (G % 3 == 0 ? 3 : 4));
    if (lanes > 8) lanes = 8;
    if (lanes > G) lanes = G;
    int rc = lanes_prepare(c, lanes);
    if (rc) return rc;
    const int la_saved = c->lookahead;
    hipStream_t const caller = c->stream;
    lanes_fork(c, lanes, caller);
    for (int g = 0; g < G && !rc; ++g) {
        SeParams p;
        rc = fill_params(&p, D, alpha[g], &rho[g], 1);
        gpmi_ctx *lc = (g % lanes == 0) ? c : c->lane[g % lanes - 1];
        if (!rc) rc = logml_core(lc, dX, n, ldx, dy, p, sigma[g] * sigma[g] + jitter, d_out3 + 3 * (size_t)g, d_info + g);
    }
    lanes_join(c, lanes, caller, la_saved);
    if (rc) return rc;
    HIPCHK(hipGetLastError());
    return 0;
}

// ARD grid: one length-scale per dimension and point (ell: G x D, point-major).  QQard's phi[[2]] is a vector
// (R/kernels.R:11-19): a grid search / optimiser over ARD length-scales evaluates exactly this.
extern "C" int gpmi_logml_grid_ard_dev(gpmi_ctx *c, const double *dX, int n, int ldx, int D, const double *dy,
                                       const double *alpha, const double *ell, const double *sigma, int G, double jitter,
                                       double *d_out3, int *d_info)
{
    ENTER(c);
    if (G < 0) return gpmi_fail(GPMI_EARG, "negative grid size");
    if (G == 0) return 0;
    if (n <= 0 || !dX || !dy || !alpha || !ell || !sigma || !d_out3 || !d_info || ldx < n)
        return gpmi_fail(GPMI_EARG, "bad argument");
    int rc;
    std::vector<SeParams> ps(G);
    for (int g = 0; g < G; ++g)
        if ((rc = fill_params(&ps[g], D, alpha[g], ell + (size_t)g * D, D))) return rc;
    if (small_logml(c, n, D, G)) {
        return small_grid(c, dX, n, ldx, D, dy, alpha, ell, D, sigma, G, jitter, d_out3, d_info);
    }
    int lanes = c->grid_lanes > 0 ? c->grid_lanes : (G % 4 == 0 ? 4 : (G % 3 == 0 ? 3 : 4));
    if (lanes > 8) lanes = 8;
    if (lanes > G) lanes = G;
    if ((rc = lanes_prepare(c, lanes))) return rc;
    const int la_saved = c->lookahead;
    hipStream_t const caller = c->stream;
    lanes_fork(c, lanes, caller);
    for (int g = 0; g < G && !rc; ++g) {
        gpmi_ctx *lc = (g % lanes == 0) ? c : c->lane[g % lanes - 1];
        rc = logml_core(lc, dX, n, ldx, dy, ps[g], sigma[g] * sigma[g] + jitter, d_out3 + 3 * (size_t)g, d_info + g);
    }
    lanes_join(c, lanes, caller, la_saved);
    if (rc) return rc;
    HIPCHK(hipGetLastError());
    return 0;
}

static int upload_xy(gpmi_ctx *c, const double *X, int n, int ldx, int D, const double *y, double **dX, double **dy)
{
    int rc;
    if ((rc = stage_buf(c, 0, (size_t)n * D * sizeof(double), dX))) return rc;
    if ((rc = stage_buf(c, 1, (size_t)n * sizeof(double), dy))) return rc;
    if ((rc = h2d_matrix(c, X, n, D, ldx, *dX))) return rc;
    HIPCHK(hipMemcpyAsync(*dy, y, (size_t)n * sizeof(double), hipMemcpyHostToDevice, c->stream));
    return 0;
}

extern "C" int gpmi_logml(gpmi_ctx *c, const double *X, int n, int ldx, int D, const double *y,
                          double alpha, const double *ell, int n_ell, double sigma, double jitter, double *out3)
{
    ENTER(c);
    if (n <= 0 || !X || !y || !out3 || ldx < n || D < 1) return gpmi_fail(GPMI_EARG, "bad argument");
    double *dX, *dy;
    int rc;
    if (small_logml(c, n, D)) {
        // Small n, host buffers: X, y are copied (by the CPU) into a pinned, device-mapped buffer that the kernel
        // reads directly, and the kernel writes (logml, sum log L_ii, z'z, info) straight into it: ONE launch and one
        // stream synchronisation, no copy call in either direction (the four of the general path cost ~55 us,
        // more than the kernel itself)
        SeParams p;
        if ((rc = fill_params(&p, D, alpha, ell, n_ell))) return rc;
        const size_t need = (8 + (size_t)n * (D + 1)) * sizeof(double);
        if ((rc = pin_reserve(c, need))) return rc;
        double *hX = c->h_pin + 8, *hy = hX + (size_t)n * D, *stage;
        for (int d = 0; d < D; ++d) memcpy(hX + (size_t)d * n, X + (size_t)d * ldx, (size_t)n * sizeof(double));
        memcpy(hy, y, (size_t)n * sizeof(double));
        if ((rc = scratch_buf(c, (size_t)n * (D + 1) * sizeof(double), &stage))) return rc;
        if ((rc = reserve_ws_small(c, n, 1))) return rc;
        double *pd = c->h_pin_dev;
        const int seq = ++c->pin_seq;
        launch_logml_small(c->stream, pd + 8, n, n, pd + 8 + (size_t)n * D, p, sigma * sigma + jitter, c->W, (size_t)c->ld, pd,
                           (int *)(pd + 3), c->d_info, stage, (int *)(pd + 7), seq);
        HIPCHK(hipGetLastError());
        if ((rc = pin_wait(c, (const int *)(c->h_pin + 7), seq))) return rc;
        memcpy(out3, c->h_pin, 3 * sizeof(double));
        return *(const int *)(c->h_pin + 3);
    }
    if ((rc = upload_xy(c, X, n, ldx, D, y, &dX, &dy))) return rc;
    if ((rc = gpmi_logml_dev(c, dX, n, n, D, dy, alpha, ell, n_ell, sigma, jitter, c->d_out, c->d_info + 1))) return rc;
    int info = 0;
    HIPCHK(hipMemcpyAsync(out3, c->d_out, 3 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipMemcpyAsync(&info, c->d_info + 1, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    if (c->timing) {
        float ms;
        for (int i = 0; i < 3; ++i) {
            c->last_ms[i] = (hipEventElapsedTime(&ms, c->ev[i], c->ev[i + 1]) == hipSuccess) ? ms : 0.0;
        }
    }
    return info;
}

extern "C" int gpmi_logml_grid(gpmi_ctx *c, const double *X, int n, int ldx, int D, const double *y,
                               const double *alpha, const double *rho, const double *sigma, int G,
                               double jitter, double *out3, int *info)
{
    ENTER(c);
    if (G < 0) return gpmi_fail(GPMI_EARG, "negative grid size");
    if (G == 0) return 0;
    if (n <= 0 || !X || !y || !out3 || !info || ldx < n || D < 1) return gpmi_fail(GPMI_EARG, "bad argument");
    double *dX, *dy, *dres;
    int rc;
    if ((rc = upload_xy(c, X, n, ldx, D, y, &dX, &dy))) return rc;
    if ((rc = scratch_buf(c, (size_t)G * (3 * sizeof(double) + sizeof(int)) + 64, &dres))) return rc;
    int *dinfo = (int *)(dres + 3 * (size_t)G);
    if ((rc = gpmi_logml_grid_dev(c, dX, n, n, D, dy, alpha, rho, sigma, G, jitter, dres, dinfo))) return rc;
    HIPCHK(hipMemcpyAsync(out3, dres, (size_t)G * 3 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipMemcpyAsync(info, dinfo, (size_t)G * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return 0;
}

extern "C" int gpmi_logml_grid_ard(gpmi_ctx *c, const double *X, int n, int ldx, int D, const double *y, const double *alpha,
                                   const double *ell, const double *sigma, int G, double jitter, double *out3, int *info)
{
    ENTER(c);
    if (G < 0) return gpmi_fail(GPMI_EARG, "negative grid size");
    if (G == 0) return 0;
    if (n <= 0 || !X || !y || !out3 || !info || ldx < n || D < 1) return gpmi_fail(GPMI_EARG, "bad argument");
    double *dX, *dy, *dres;
    int rc;
    if ((rc = upload_xy(c, X, n, ldx, D, y, &dX, &dy))) return rc;
    if ((rc = scratch_buf(c, (size_t)G * (3 * sizeof(double) + sizeof(int)) + 64, &dres))) return rc;
    int *dinfo = (int *)(dres + 3 * (size_t)G);
    if ((rc = gpmi_logml_grid_ard_dev(c, dX, n, n, D, dy, alpha, ell, sigma, G, jitter, dres, dinfo))) return rc;
    HIPCHK(hipMemcpyAsync(out3, dres, (size_t)G * 3 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipMemcpyAsync(info, dinfo, (size_t)G * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return 0;
}

static int joint_logml_core(gpmi_ctx *c, const double *dt, int n, const double *dyy, double alpha, double l, double sigma,
                            double jitter, double *d_out3, int *d_info)
{
    int rc;
    const int n2 = 2 * n, M = n2 + 1;
    if ((rc = reserve_ws(c, M, n2))) return rc;
    const size_t ld = (size_t)c->ld;
    tic(c, 0);
    HIPCHK(hipMemsetAsync(c->d_info, 0, sizeof(int), c->stream));
    kt_begin(c, 0);
    launch_joint_cov(c->stream, dt, n, alpha * alpha, l, sigma * sigma, jitter, 0, 1, c->W, ld);
    kt_end(c, 0, 4.0 * (double)n2 * ((double)n2 + 1.0));
    launch_set_row(c->stream, c->W, ld, n2, dyy, n2, n2);
    tic(c, 1);
    if ((rc = launch_potrf_partial(c, c->W, ld, M, n2, n2, c->d_info, nullptr))) return rc;
    tic(c, 2);
    launch_logml_finalize(c->stream, c->W, ld, n2, n2, c->d_info, d_out3, d_info, c->d_fin);
    tic(c, 3);
    HIPCHK(hipGetLastError());
    return 0;
}

extern "C" int gpmi_joint_logml_dev(gpmi_ctx *c, const double *dt, int n, const double *dyy, double alpha,
                                    double l, double sigma, double jitter, double *d_out3, int *d_info)
{
    ENTER(c);
    if (n <= 0 || !dt || !dyy || !d_out3 || !(l > 0.0)) return gpmi_fail(GPMI_EARG, "bad argument");
    return joint_logml_core(c, dt, n, dyy, alpha, l, sigma, jitter, d_out3, d_info);
}

// G independent (alpha, l, sigma) points of the joint [y, y'] model on the same data, on the lanes
extern "C" int gpmi_joint_logml_grid_dev(gpmi_ctx *c, const double *dt, int n, const double *dyy, const double *alpha,
                                         const double *l, const double *sigma, int G, double jitter, double *d_out3, int *d_info)
{
    ENTER(c);
    if (G < 0) return gpmi_fail(GPMI_EARG, "negative grid size");
    if (G == 0) return 0;
    if (n <= 0 || !dt || !dyy || !alpha || !l || !sigma || !d_out3 || !d_info) return gpmi_fail(GPMI_EARG, "bad argument");
    for (int g = 0; g < G; ++g)
        if (!(l[g] > 0.0)) return gpmi_fail(GPMI_EARG, "length-scale must be positive");
    int lanes = c->grid_lanes > 0 ? c->grid_lanes : (G % 4 == 0 ? 4 : (G % 3 == 0 ? 3 : 4));
    if (lanes > 8) lanes = 8;
    if (lanes > G) lanes = G;
    int rc = lanes_prepare(c, lanes);
    if (rc) return rc;
    for (int k = 0; k < lanes; ++k)
        if ((rc = reserve_ws(k ? c->lane[k - 1] : c, 2 * n + 1, 2 * n))) return rc;
    const int la_saved = c->lookahead;
    hipStream_t const caller = c->stream;
    lanes_fork(c, lanes, caller);
    for (int g = 0; g < G && !rc; ++g) {
        gpmi_ctx *lc = (g % lanes == 0) ? c : c->lane[g % lanes - 1];
        rc = joint_logml_core(lc, dt, n, dyy, alpha[g], l[g], sigma[g], jitter, d_out3 + 3 * (size_t)g, d_info + g);
    }
    lanes_join(c, lanes, caller, la_saved);
    if (rc) return rc;
    HIPCHK(hipGetLastError());
    return 0;
}

extern "C" int gpmi_joint_logml(gpmi_ctx *c, const double *t, int n, const double *yy, double alpha,
                                double l, double sigma, double jitter, double *out3)
{
    ENTER(c);
    if (n <= 0 || !t || !yy || !out3) return gpmi_fail(GPMI_EARG, "bad argument");
    double *dt, *dyy;
    int rc;
    if ((rc = stage_buf(c, 0, (size_t)n * sizeof(double), &dt))) return rc;
    if ((rc = stage_buf(c, 1, (size_t)2 * n * sizeof(double), &dyy))) return rc;
    HIPCHK(hipMemcpyAsync(dt, t, (size_t)n * sizeof(double), hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(dyy, yy, (size_t)2 * n * sizeof(double), hipMemcpyHostToDevice, c->stream));
    if ((rc = gpmi_joint_logml_dev(c, dt, n, dyy, alpha, l, sigma, jitter, c->d_out, c->d_info + 1))) return rc;
    int info = 0;
    HIPCHK(hipMemcpyAsync(out3, c->d_out, 3 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipMemcpyAsync(&info, c->d_info + 1, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return info;
}

// ---- rbf_cov_chol (covariance.cpp:9-47) ---------------------------------------
namespace {
__global__ void k_rbf_dsigma(const double *__restrict__ x, int n, double l, double *__restrict__ S, size_t ld)
{
    // dSigma/dl = Sigma * (xi-xj)^2 / l^3  (tangent of covariance.cpp:19 with dl = 1, :13)
    const int i = blockIdx.x * 64 + (threadIdx.x & 63);
    const int j0 = blockIdx.y * 16 + (threadIdx.x >> 6) * 4;
    if (i >= n) return;
    const double xi = x[i];
    for (int q = 0; q < 4; ++q) {
        const int j = j0 + q;
        if (j >= n) break;
        const double r = xi - x[j], r2 = r * r;
        S[(size_t)i + (size_t)j * ld] = exp(-r2 / (2 * l * l)) * r2 / (l * l * l);
    }
}
}  // namespace

// every buffer rbf_cov_chol_core(c, ., n, ...) uses, at its final size
static int rbf_cov_chol_reserve(gpmi_ctx *c, int n)
{
    int rc;
    if ((rc = reserve_ws(c, n, n))) return rc;
    const int ldd = ((n + 15) / 16) * 16 + 16;
    const size_t msz = (size_t)ldd * (n + 1) * sizeof(double);
    const int npan = (n + GPMI_NB - 1) / GPMI_NB;
    double *b;
    for (int slot = 1; slot <= 3; ++slot)
        if ((rc = stage_buf(c, slot, msz, &b))) return rc;
    return scratch_buf(c, (size_t)npan * GPMI_FPACK * sizeof(double), &b);
}

// device core: x resident in dx; on return (stream order) Lc holds L (zero upper) and S holds dL/dl,
// both n x n with leading dimension ldd in the context's staging buffers 3 and 1
static int rbf_cov_chol_core(gpmi_ctx *c, const double *dx, int n, double l, double **Lc_out, double **S_out, int *ldd_out)
{
    int rc;
    if ((rc = reserve_ws(c, n, n))) return rc;
    const size_t ld = (size_t)c->ld;
    const int ldd = ((n + 15) / 16) * 16 + 16;
    const size_t msz = (size_t)ldd * (n + 1) * sizeof(double);
    const int npan = (n + GPMI_NB - 1) / GPMI_NB;
    double *S, *S2, *Lc, *Fall;
    if ((rc = stage_buf(c, 1, msz, &S))) return rc;
    if ((rc = stage_buf(c, 2, msz, &S2))) return rc;
    if ((rc = stage_buf(c, 3, msz, &Lc))) return rc;
    if ((rc = scratch_buf(c, (size_t)npan * GPMI_FPACK * sizeof(double), &Fall))) return rc;
    hipStream_t s = c->stream;
    HIPCHK(hipMemsetAsync(c->d_info, 0, sizeof(int), s));
    // Sigma (+1e-10 jitter, covariance.cpp:23-25) -> L
    SeParams p;
    double ell = l;
    if ((rc = fill_params(&p, 1, 1.0, &ell, 1))) return rc;
    launch_se_cov(c, s, dx, n, n, nullptr, n, n, p, 1e-10, 1, c->W, ld);
    if ((rc = launch_potrf_partial(c, c->W, ld, n, n, n, c->d_info, Fall))) return rc;
    launch_copy_matrix(s, c->W, ld, Lc, (size_t)ldd, n, n, 1);
    // tangent: Ldot = L Phi(L^-1 Sdot L^-T), Phi = lower triangle with halved diagonal
    hipLaunchKernelGGL(k_rbf_dsigma, dim3((n + 63) / 64, (n + 15) / 16), 256, 0, s, dx, n, l, S, (size_t)ldd);
    if ((rc = launch_trsm_right(c, Lc, (size_t)ldd, n, S, (size_t)ldd, n, Fall))) return rc;     // S <- Sdot L^-T
    launch_transpose(s, S, (size_t)ldd, S2, (size_t)ldd, n, n);                                  // S2 = L^-1 Sdot
    // M = L^-1 Sdot L^-T is symmetric and only Phi(M) -- its lower triangle -- is used: the second solve computes
    // the lower triangle alone (n^3 / 3 flops instead of n^3), and L Phi, a product of two lower-triangular
    // matrices, only its lower tiles over the K range where both operands are non-zero (n^3 / 3 instead of 2 n^3)
    if ((rc = launch_trsm_right(c, Lc, (size_t)ldd, n, S2, (size_t)ldd, n, Fall, 2))) return rc; // lower(S2) = lower(M)
    launch_transpose(s, S2, (size_t)ldd, S, (size_t)ldd, n, n);                                   // upper(S) = lower(M)^T
    launch_phi_mask(s, S, (size_t)ldd, n);                                                        // S = Phi^T (upper)
    HIPCHK(hipMemsetAsync(S2, 0, (size_t)ldd * n * sizeof(double), s));                           // tiles above the diagonal
    launch_gemm_tri_lower(s, Lc, (size_t)ldd, S, (size_t)ldd, S2, (size_t)ldd, n);                // S2 = L Phi
    HIPCHK(hipGetLastError());
    *Lc_out = Lc;
    *S_out = S2;
    *ldd_out = ldd;
    return 0;
}

extern "C" int gpmi_rbf_cov_chol(gpmi_ctx *c, const double *x, int n, double l, double *L, int ldl,
                                 double *dLdl, int lddl)
{
    ENTER(c);
    if (n <= 0 || !x || !L || !dLdl || ldl < n || lddl < n || !(l > 0.0)) return gpmi_fail(GPMI_EARG, "bad argument");
    int rc, ldd;
    double *dx, *Lc, *S;
    hipStream_t s = c->stream;
    if (n <= 64) {
        // small n: ONE launch of one workgroup; x in, L / dL/dl / info out through the pinned, device-mapped buffer:
        // [info | x | L (n x n) | dLdl (n x n)]  (tools/interp_small_bench.py: n = 50 144 -> 84 us; at n = 100 one workgroup
        // takes 170 us against the chain's 157 -- ONE call stays on the chain there, the table build below does not)
        const size_t o_x = 8, o_L = o_x + n, o_dL = o_L + (size_t)n * n;
        if ((rc = pin_reserve(c, (o_dL + (size_t)n * n) * sizeof(double)))) return rc;
        memcpy(c->h_pin + o_x, x, (size_t)n * sizeof(double));
        double *stage;
        if ((rc = scratch_buf(c, (size_t)n * sizeof(double), &stage))) return rc;
        if ((rc = reserve_ws_small(c, n, 3))) return rc;
        double *pd = c->h_pin_dev;
        launch_rbf_cov_chol_small(s, pd + o_x, n, &l, 1, c->W, pd + o_L, pd + o_dL, 0, (size_t)n, (int *)pd, c->d_info, stage);
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(s));
        for (int j = 0; j < n; ++j) {
            memcpy(L + (size_t)j * ldl, c->h_pin + o_L + (size_t)j * n, (size_t)n * sizeof(double));
            memcpy(dLdl + (size_t)j * lddl, c->h_pin + o_dL + (size_t)j * n, (size_t)n * sizeof(double));
        }
        return *(const int *)c->h_pin;
    }
    if ((rc = stage_buf(c, 0, (size_t)n * sizeof(double), &dx))) return rc;
    HIPCHK(hipMemcpyAsync(dx, x, (size_t)n * sizeof(double), hipMemcpyHostToDevice, s));
    if ((rc = rbf_cov_chol_core(c, dx, n, l, &Lc, &S, &ldd))) return rc;
    int info = 0;
    HIPCHK(hipMemcpyAsync(&info, c->d_info, sizeof(int), hipMemcpyDeviceToHost, s));
    if ((rc = d2h_matrix(c, Lc, (size_t)ldd, n, n, L, ldl))) return rc;
    if ((rc = d2h_matrix(c, S, (size_t)ldd, n, n, dLdl, lddl))) return rc;
    HIPCHK(hipStreamSynchronize(s));
    return info;
}

// ---- Cholesky-factor interpolation over the length-scale ---------------------------
extern "C" int gpmi_interp_free(gpmi_ctx *c)
{
    ENTER(c);
    HIPCHK(hipStreamSynchronize(c->stream));
    if (c->itp_L) (void)hipFree(c->itp_L);
    if (c->itp_dL) (void)hipFree(c->itp_dL);
    if (c->itp_part) (void)hipFree(c->itp_part);
    free(c->itp_lp);
    c->itp_L = c->itp_dL = c->itp_part = nullptr;
    c->itp_lp = nullptr;
    c->itp_P = c->itp_n = 0;
    return 0;
}

static int interp_alloc(gpmi_ctx *c, const double *lp, int P, int n)
{
    if (P < 2 || n <= 0 || !lp) return gpmi_fail(GPMI_EARG, "interpolation table needs P >= 2 length-scales");
    for (int p = 0; p + 1 < P; ++p)
        if (!(lp[p + 1] > lp[p])) return gpmi_fail(GPMI_EARG, "lp must be strictly increasing");
    int rc = gpmi_interp_free(c);
    if (rc) return rc;
    c->itp_ld = (size_t)((n + 1) & ~1);
    const size_t bytes = (size_t)P * c->itp_ld * n * sizeof(double);
    if (hipMalloc((void **)&c->itp_L, bytes) != hipSuccess || hipMalloc((void **)&c->itp_dL, bytes) != hipSuccess ||
        hipMalloc((void **)&c->itp_part, 2 * (size_t)hermite_mv_chunks(n) * n * sizeof(double)) != hipSuccess) {
        (void)hipGetLastError();
        gpmi_interp_free(c);
        return gpmi_fail(GPMI_ENOMEM, "interpolation table of %d x %d x %d does not fit", P, n, n);
    }
    c->itp_lp = (double *)malloc(sizeof(double) * P);
    if (!c->itp_lp) return gpmi_fail(GPMI_ENOMEM, "host allocation failed");
    memcpy(c->itp_lp, lp, sizeof(double) * P);
    c->itp_P = P;
    c->itp_n = n;
    return 0;
}

extern "C" int gpmi_interp_build(gpmi_ctx *c, const double *x, int n, const double *lp, int P)
{
    ENTER(c);
    if (!x) return gpmi_fail(GPMI_EARG, "bad argument");
    int rc;
    if ((rc = interp_alloc(c, lp, P, n))) return rc;
    for (int p = 0; p < P; ++p)
        if (!(lp[p] > 0.0)) return gpmi_fail(GPMI_EARG, "length-scales must be positive");
    double *dx;
    int *dinfo;
    if ((rc = stage_buf(c, 0, (size_t)n * sizeof(double) + (size_t)P * sizeof(int) + 64, &dx))) return rc;
    dinfo = (int *)(dx + n + 1);
    if (n <= GPMI_NB) {
        // the reference's size (N = 100, P = 10): the whole table in launches of up to 64 workgroups, one entry each, written
        // straight into the table
        hipStream_t s = c->stream;
        HIPCHK(hipMemcpyAsync(dx, x, (size_t)n * sizeof(double), hipMemcpyHostToDevice, s));
        const int per = P < 64 ? P : 64;
        if ((rc = reserve_ws_small(c, n, 3 * per))) return rc;
        if ((rc = reserve_small_par(c, per))) return rc;
        const size_t msz = c->itp_ld * (size_t)n;
        for (int p0 = 0; p0 < P; p0 += per) {
            const int pc = (P - p0 < per) ? P - p0 : per;
            launch_rbf_cov_chol_small(s, dx, n, lp + p0, pc, c->W, c->itp_L + (size_t)p0 * msz, c->itp_dL + (size_t)p0 * msz, msz,
                                      c->itp_ld, dinfo + p0, c->d_sinfo, nullptr);
        }
        std::vector<int> info(P, 0);
        HIPCHK(hipMemcpyAsync(info.data(), dinfo, (size_t)P * sizeof(int), hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        HIPCHK(hipGetLastError());
        for (int p = 0; p < P; ++p)
            if (info[p]) return info[p];
        return 0;
    }
    // The P table entries (test_interpolate.R:9-19: P calls of rbf_cov_chol; interpolated_gp.stan:10-28)
    // are independent factorisations: they run on the grid lanes, entry p on lane p mod lanes, every
    // lane in its own workspace and staging buffers, and nothing synchronises with the host until
    // all are enqueued.
    int lanes = c->grid_lanes > 0 ? c->grid_lanes : 4;
    if (lanes > 8) lanes = 8;
    if (lanes > P) lanes = P;
    if ((rc = lanes_prepare(c, lanes))) return rc;
    // buffers every lane needs, sized BEFORE the fork (a growing buffer synchronises its stream)
    for (int l = 0; l < lanes; ++l) {
        gpmi_ctx *lc = l ? c->lane[l - 1] : c;
        if ((rc = rbf_cov_chol_reserve(lc, n))) return rc;
    }
    const int la_saved = c->lookahead;
    hipStream_t const caller = c->stream;
    HIPCHK(hipMemcpyAsync(dx, x, (size_t)n * sizeof(double), hipMemcpyHostToDevice, caller));
    lanes_fork(c, lanes, caller);
    for (int p = 0; p < P && !rc; ++p) {
        gpmi_ctx *lc = (p % lanes == 0) ? c : c->lane[p % lanes - 1];
        double *Lc, *S;
        int ldd;
        rc = rbf_cov_chol_core(lc, dx, n, lp[p], &Lc, &S, &ldd);
        if (rc) break;
        hipStream_t s = lc->stream;
        launch_copy_matrix(s, Lc, (size_t)ldd, c->itp_L + (size_t)p * c->itp_ld * n, c->itp_ld, n, n, 0);
        launch_copy_matrix(s, S, (size_t)ldd, c->itp_dL + (size_t)p * c->itp_ld * n, c->itp_ld, n, n, 0);
        if (hipMemcpyAsync(dinfo + p, lc->d_info, sizeof(int), hipMemcpyDeviceToDevice, s) != hipSuccess)
            rc = gpmi_fail(GPMI_EHIP, "info copy failed");
    }
    lanes_join(c, lanes, caller, la_saved);
    if (rc) return rc;
    std::vector<int> info(P, 0);
    HIPCHK(hipMemcpyAsync(info.data(), dinfo, (size_t)P * sizeof(int), hipMemcpyDeviceToHost, caller));
    HIPCHK(hipStreamSynchronize(caller));
    HIPCHK(hipGetLastError());
    for (int p = 0; p < P; ++p)
        if (info[p]) return info[p];
    return 0;
}

extern "C" int gpmi_interp_load(gpmi_ctx *c, const double *lp, int P, const double *Ls, const double *dLdls,
                                int n, int ld)
{
    ENTER(c);
    if (!Ls || !dLdls || ld < n) return gpmi_fail(GPMI_EARG, "bad argument");
    int rc;
    if ((rc = interp_alloc(c, lp, P, n))) return rc;
    for (int p = 0; p < P; ++p) {
        HIPCHK(hipMemcpy2DAsync(c->itp_L + (size_t)p * c->itp_ld * n, c->itp_ld * sizeof(double),
                                Ls + (size_t)p * ld * n, (size_t)ld * sizeof(double), (size_t)n * sizeof(double), n,
                                hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipMemcpy2DAsync(c->itp_dL + (size_t)p * c->itp_ld * n, c->itp_ld * sizeof(double),
                                dLdls + (size_t)p * ld * n, (size_t)ld * sizeof(double), (size_t)n * sizeof(double), n,
                                hipMemcpyHostToDevice, c->stream));
    }
    HIPCHK(hipStreamSynchronize(c->stream));
    return 0;
}

// interval of the reference (covariance.cpp:57-61), clamped to the last one
static int interp_interval(const gpmi_ctx *c, double l)
{
    int lidx = 0;
    for (; lidx < c->itp_P - 1; ++lidx)
        if (c->itp_lp[lidx + 1] >= l) break;
    return lidx > c->itp_P - 2 ? c->itp_P - 2 : lidx;
}

extern "C" int gpmi_approx_L(gpmi_ctx *c, double l, double *out, int ldo)
{
    ENTER(c);
    if (!c->itp_L) return gpmi_fail(GPMI_EARG, "no interpolation table (gpmi_interp_build / gpmi_interp_load first)");
    const int n = c->itp_n;
    if (!out || ldo < n || !(l == l)) return gpmi_fail(GPMI_EARG, "bad argument");
    int rc;
    double *dout;
    const size_t ldd = (size_t)((n + 1) & ~1);
    if ((rc = stage_buf(c, 1, ldd * n * sizeof(double), &dout))) return rc;
    const int k = interp_interval(c, l);
    const size_t msz = c->itp_ld * n;
    launch_hermite_blend(c->stream, c->itp_L + k * msz, c->itp_L + (k + 1) * msz, c->itp_dL + k * msz,
                         c->itp_dL + (k + 1) * msz, c->itp_ld, n, c->itp_lp[k], c->itp_lp[k + 1], l, dout, ldd);
    if ((rc = d2h_matrix(c, dout, ldd, n, n, out, ldo))) return rc;
    HIPCHK(hipStreamSynchronize(c->stream));
    return 0;
}

static int approx_Lz_core(gpmi_ctx *c, double l, const double *dz, double *df, double *ddfdl)
{
    if (!c->itp_L) return gpmi_fail(GPMI_EARG, "no interpolation table (gpmi_interp_build / gpmi_interp_load first)");
    if (!dz || !df || !(l == l)) return gpmi_fail(GPMI_EARG, "bad argument");
    const int n = c->itp_n;
    const int k = interp_interval(c, l);
    const size_t msz = c->itp_ld * n;
    if (n > GPMI_HMV_SMALL_N)   // (the one-launch form of small n keeps its chunk sums in registers)
        HIPCHK(hipMemsetAsync(c->itp_part, 0, (ddfdl ? 2 : 1) * (size_t)hermite_mv_chunks(n) * n * sizeof(double), c->stream));
    launch_hermite_mv(c->stream, c->itp_L + k * msz, c->itp_L + (k + 1) * msz, c->itp_dL + k * msz,
                      c->itp_dL + (k + 1) * msz, c->itp_ld, n, c->itp_lp[k], c->itp_lp[k + 1], l, dz, c->itp_part, df, ddfdl);
    HIPCHK(hipGetLastError());
    return 0;
}

extern "C" int gpmi_approx_Lz_dev(gpmi_ctx *c, double l, const double *dz, double *df)
{
    ENTER(c);
    return approx_Lz_core(c, l, dz, df, nullptr);
}

extern "C" int gpmi_approx_Lz_grad_dev(gpmi_ctx *c, double l, const double *dz, double *df, double *ddfdl)
{
    ENTER(c);
    if (!ddfdl) return gpmi_fail(GPMI_EARG, "bad argument");
    return approx_Lz_core(c, l, dz, df, ddfdl);
}

static int approx_Lz_host(gpmi_ctx *c, double l, const double *z, double *f, double *dfdl)
{
    if (!c->itp_L) return gpmi_fail(GPMI_EARG, "no interpolation table (gpmi_interp_build / gpmi_interp_load first)");
    if (!z || !f) return gpmi_fail(GPMI_EARG, "bad argument");
    const int n = c->itp_n;
    int rc;
    double *dz;
    if ((rc = stage_buf(c, 0, 3 * (size_t)n * sizeof(double), &dz))) return rc;
    if (n <= 4096) {
        // the per-iteration call of the interpolated model at the reference's size (N = 100, test_interpolate.R:5): z goes in
        // and f (dfdl) come back through the pinned, device-mapped buffer -- three small launches, no copy call
        const size_t need = (3 * (size_t)n + 8) * sizeof(double);
        if ((rc = pin_reserve(c, need))) return rc;
        memcpy(c->h_pin, z, (size_t)n * sizeof(double));
        double *pd = c->h_pin_dev;
        const double *zsrc = pd;
        if (n > GPMI_HMV_SMALL_N) {   // the chunked kernels read z once per row block: staged in device memory first
            launch_copy_matrix(c->stream, pd, (size_t)n, dz, (size_t)n, n, 1, 0);
            zsrc = dz;
        }
        if ((rc = approx_Lz_core(c, l, zsrc, pd + n, dfdl ? pd + 2 * (size_t)n : nullptr))) return rc;
        HIPCHK(hipStreamSynchronize(c->stream));
        memcpy(f, c->h_pin + n, (size_t)n * sizeof(double));
        if (dfdl) memcpy(dfdl, c->h_pin + 2 * (size_t)n, (size_t)n * sizeof(double));
        return 0;
    }
    HIPCHK(hipMemcpyAsync(dz, z, (size_t)n * sizeof(double), hipMemcpyHostToDevice, c->stream));
    if ((rc = approx_Lz_core(c, l, dz, dz + n, dfdl ? dz + 2 * (size_t)n : nullptr))) return rc;
    HIPCHK(hipMemcpyAsync(f, dz + n, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    if (dfdl) HIPCHK(hipMemcpyAsync(dfdl, dz + 2 * (size_t)n, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return 0;
}

extern "C" int gpmi_approx_Lz(gpmi_ctx *c, double l, const double *z, double *f)
{
    ENTER(c);
    return approx_Lz_host(c, l, z, f, nullptr);
}

extern "C" int gpmi_approx_Lz_grad(gpmi_ctx *c, double l, const double *z, double *f, double *dfdl)
{
    ENTER(c);
    if (!dfdl) return gpmi_fail(GPMI_EARG, "bad argument");
    return approx_Lz_host(c, l, z, f, dfdl);
}

// ---- gradient of the log marginal likelihood ---------------------------
// d logml / d theta = 1/2 tr((a a' - K^-1) dK/dtheta), a = K^-1 y: what Stan's reverse-mode autodiff
// hands NUTS for models/fit_hyperparameters.stan:18-32.  K^-1 = L^-T L^-1 is formed explicitly:
// L^-T by a triangular-aware panel substitution from the identity (N^3/3 flop), then -K^-1 by the
// same SYRK kernel as the factorisation's trailing update (N^3 flop) -- four Cholesky's worth of
// MFMA work on top of the factorisation; the final contraction re-evaluates the kernel from the
// inputs instead of reading K back.
namespace {
__global__ void k_set_identity(double *__restrict__ A, size_t ld, int n)
{
    const int i = blockIdx.x * 64 + (threadIdx.x & 63);
    const int j0 = blockIdx.y * 16 + (threadIdx.x >> 6) * 4;
    if (i >= n) return;
    for (int q = 0; q < 4; ++q) {
        const int j = j0 + q;
        if (j < n) A[(size_t)i + (size_t)j * ld] = (i == j) ? 1.0 : 0.0;
    }
}

// a = U z for upper-triangular U (= L^-T): thread = row, 512-column chunks on a 2-D grid (a
// one-thread-per-row loop over all columns is a latency chain: 8 ms at N = 16384), chunks wholly
// below the diagonal skipped, partial sums added in chunk order (fixed: deterministic)
constexpr int UMV_ROWS = 256, UMV_COLS = 512;
__global__ __launch_bounds__(UMV_ROWS) void k_upper_mv_part(const double *__restrict__ U, size_t ld, int n,
                                                            const double *__restrict__ z, double *__restrict__ part)
{
    const int r0 = blockIdx.x * UMV_ROWS, i = r0 + threadIdx.x;
    const int c0 = blockIdx.y * UMV_COLS;
    const int c1 = (c0 + UMV_COLS < n) ? c0 + UMV_COLS : n;
    if (i >= n) return;
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    if (c1 > r0) {  // some column j >= some row of this block
        int j = c0 > i ? c0 : i;
        const double *col = U + (size_t)i + (size_t)j * ld;
        for (; j + 4 <= c1; j += 4) {
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[q] = fma(col[(size_t)q * ld], z[j + q], acc[q]);
            col += 4 * ld;
        }
        for (; j < c1; ++j) {
            acc[0] = fma(col[0], z[j], acc[0]);
            col += ld;
        }
    }
    part[(size_t)blockIdx.y * n + i] = (acc[0] + acc[1]) + (acc[2] + acc[3]);
}

__global__ __launch_bounds__(256) void k_upper_mv_sum(const double *__restrict__ part, int n, int nchunk,
                                                      double *__restrict__ a)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    double acc = 0.0;
    for (int q = 0; q < nchunk; ++q) acc += part[(size_t)q * n + i];
    a[i] = acc;
}

// per 64 x 64 tile of the lower triangle: sums of w g kse, w g kse (x_id - x_jd)^2 (d < D), [i == j] g
// with g = (a_i a_j + Wij) / 2 (W holds -K^-1), w = 2 off the diagonal (symmetry), 1 on it
constexpr int GRAD_NS = 2 + GPMI_MAXD;           // slots per tile of the D <= 8 kernel
constexpr int GRAD_NS_MAX = 2 + GPMI_MAXD_BIG;   // ... and the most the generic one writes (D = 64)
static inline int grad_ns(int D) { return 2 + ((D + 7) / 8) * 8; }
__global__ __launch_bounds__(256) void k_grad_partial(const double *__restrict__ X, int n, int ldx, SeParams p,
                                                      const double *__restrict__ a, const double *__restrict__ W,
                                                      size_t ld, double *__restrict__ part)
{
    __shared__ double red[256];
    const int ti = blockIdx.x, tj = blockIdx.y;
    if (tj > ti) return;
    const int i = ti * 64 + (threadIdx.x & 63);
    const int jbase = tj * 64 + (threadIdx.x >> 6) * 16;
    double acc[GRAD_NS];
#pragma unroll
    for (int s = 0; s < GRAD_NS; ++s) acc[s] = 0.0;
    if (i < n) {
        double xi[GPMI_MAXD];
#pragma unroll
        for (int d = 0; d < GPMI_MAXD; ++d) xi[d] = d < p.D ? X[(size_t)i + (size_t)d * ldx] : 0.0;
        const double ai = a[i];
        for (int q = 0; q < 16; ++q) {
            const int j = jbase + q;
            if (j >= n || j > i) break;
            double e = 0.0, r2[GPMI_MAXD];
#pragma unroll
            for (int d = 0; d < GPMI_MAXD; ++d) {
                const double r = d < p.D ? xi[d] - X[(size_t)j + (size_t)d * ldx] : 0.0;
                r2[d] = r * r;
                e += r2[d] * p.inv_ell[d] * p.inv_ell[d];
            }
            const double kse = p.a2 * exp(-0.5 * e);
            const double g = 0.5 * (ai * a[j] + W[(size_t)i + (size_t)j * ld]);
            const double w = (i == j) ? 1.0 : 2.0;
            acc[0] += w * g * kse;
#pragma unroll
            for (int d = 0; d < GPMI_MAXD; ++d) acc[1 + d] += w * g * kse * r2[d];
            if (i == j) acc[1 + GPMI_MAXD] += g;
        }
    }
    const size_t slot = ((size_t)ti * (ti + 1) / 2 + tj) * GRAD_NS;
    for (int s = 0; s < GRAD_NS; ++s) {  // fixed-shape tree per quantity: deterministic
        red[threadIdx.x] = acc[s];
        __syncthreads();
        for (int h = 128; h > 0; h >>= 1) {
            if ((int)threadIdx.x < h) red[threadIdx.x] += red[threadIdx.x + h];
            __syncthreads();
        }
        if (threadIdx.x == 0) part[slot + s] = red[0];
        __syncthreads();
    }
}

// The same contraction for ANY D <= GPMI_MAXD_BIG (QQard takes any D, R/kernels.R:11-19): 64 x 64 tile of the lower
// triangle, thread = one row x 16 columns.  The coordinates of the tile's rows and columns are staged in LDS in
// chunks of 16 dimensions; pass 1 accumulates the scaled squared distance of the thread's 16 pairs over all D and
// turns it into c_q = w g kse (16 registers), pass 2 walks the dimensions again and reduces sum_q c_q (x_id - x_jd)^2
// over the workgroup, one fixed-shape tree per dimension (deterministic).  Slots per tile: ns = 2 + roundup(D, 8):
// [0] sum c, [1 + d] per dimension, [ns - 1] the diagonal term -- the layout of k_grad_partial for D <= 8.
// O(N^2 D) next to the O(N^3) of K^-1: 64 trees per tile at D = 64 are noise.
__global__ __launch_bounds__(256) void k_grad_partial_big(const double *__restrict__ X, int n, int ldx, SeParams p,
                                                          const double *__restrict__ a, const double *__restrict__ W,
                                                          size_t ld, double *__restrict__ part, int ns)
{
    __shared__ double sx[16][64], sy[16][64], red[256];
    const int ti = blockIdx.x, tj = blockIdx.y;
    if (tj > ti) return;
    const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int i = ti * 64 + lane, j0 = tj * 64 + grp * 16;
    const int ic = i < n ? i : n - 1;
    double e[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) e[q] = 0.0;
    auto stage = [&](int d0, int dc) {
        __syncthreads();
        for (int k = grp; k < 2 * dc; k += 4) {   // wave-uniform: operand and dimension
            const int dd = k >> 1;
            if (k & 1) {
                const int j = tj * 64 + lane;
                sy[dd][lane] = X[(size_t)(j < n ? j : n - 1) + (size_t)(d0 + dd) * ldx];
            } else {
                sx[dd][lane] = X[(size_t)ic + (size_t)(d0 + dd) * ldx];
            }
        }
        __syncthreads();
    };
    for (int d0 = 0; d0 < p.D; d0 += 16) {
        const int dc = p.D - d0 < 16 ? p.D - d0 : 16;
        stage(d0, dc);
        for (int dd = 0; dd < dc; ++dd) {
            const double xi = sx[dd][lane], ie2 = p.inv_ell[d0 + dd] * p.inv_ell[d0 + dd];
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const double r = xi - sy[dd][grp * 16 + q];
                e[q] += r * r * ie2;
            }
        }
    }
    double c0 = 0.0, cdiag = 0.0;
#pragma unroll
    for (int q = 0; q < 16; ++q) {   // e[q] becomes c_q
        const int j = j0 + q;
        double cq = 0.0;
        if (i < n && j < n && j <= i) {
            const double g = 0.5 * (a[i] * a[j] + W[(size_t)i + (size_t)j * ld]);
            cq = ((i == j) ? 1.0 : 2.0) * g * (p.a2 * exp(-0.5 * e[q]));
            if (i == j) cdiag += g;
        }
        e[q] = cq;
        c0 += cq;
    }
    const size_t slot = ((size_t)ti * (ti + 1) / 2 + tj) * (size_t)ns;
    auto reduce_to = [&](double v, size_t dst) {
        red[threadIdx.x] = v;
        __syncthreads();
        for (int h = 128; h > 0; h >>= 1) {
            if ((int)threadIdx.x < h) red[threadIdx.x] += red[threadIdx.x + h];
            __syncthreads();
        }
        if (threadIdx.x == 0) part[dst] = red[0];
        __syncthreads();
    };
    reduce_to(c0, slot);
    reduce_to(cdiag, slot + (size_t)ns - 1);
    for (int d0 = 0; d0 < p.D; d0 += 16) {
        const int dc = p.D - d0 < 16 ? p.D - d0 : 16;
        stage(d0, dc);
        for (int dd = 0; dd < dc; ++dd) {
            const double xi = sx[dd][lane];
            double v = 0.0;
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const double r = xi - sy[dd][grp * 16 + q];
                v += e[q] * (r * r);
            }
            reduce_to(v, slot + 1 + (size_t)(d0 + dd));
        }
    }
}

__global__ __launch_bounds__(1024) void k_grad_final(const double *__restrict__ part, size_t ntiles,
                                                     double *__restrict__ sums, int ns)
{
    __shared__ double red[1024];
    for (int s = 0; s < ns; ++s) {
        double acc = 0.0;
        for (size_t t = threadIdx.x; t < ntiles; t += 1024) acc += part[t * (size_t)ns + s];
        red[threadIdx.x] = acc;
        __syncthreads();
        for (int h = 512; h > 0; h >>= 1) {
            if ((int)threadIdx.x < h) red[threadIdx.x] += red[threadIdx.x + h];
            __syncthreads();
        }
        if (threadIdx.x == 0) sums[s] = red[0];
        __syncthreads();
    }
}
}  // namespace

// Mid sizes (n <= tune.grad_aug_n): K^-1 and a = K^-1 y from ONE augmented partial factorisation instead of a second and a
// third launch chain (L^-T from the identity, then U U^T).  The workspace holds the (2n + 1)-row matrix
//     [ K        .   ]   n rows          after factoring the first n columns (launch_potrf_partial, the kernel chain of
//     [ y^T      0   ]   1 row           gpmi_gp_condition) the rows below are [z^T; U], U = I L^-T = L^-T, and the
//     [ I        0   ]   n rows          trailing block is 0 - [z^T; U][z^T; U]^T = -[[z'z, a^T], [a, K^-1]],  a = U z.
// The triangular structure of I and U is not exploited (2.3 n^3 flops instead of n^3) -- at these sizes the evaluation is a
// latency chain of ~36 us per 128-column panel, and this form has ONE chain where the other has three.
namespace {
__global__ void k_aug_identity(double *__restrict__ A, size_t ld, int n)   // A: rows n + 1 .. 2n of columns 0 .. n - 1
{
    const int i = blockIdx.x * 64 + (threadIdx.x & 63);
    const int j0 = blockIdx.y * 16 + (threadIdx.x >> 6) * 4;
    if (i >= n) return;
    for (int q = 0; q < 4; ++q) {
        const int j = j0 + q;
        if (j < n) A[(size_t)i + (size_t)j * ld] = (i == j) ? 1.0 : 0.0;
    }
}
}  // namespace

// many: several evaluations share the chip (the lanes): flops count for more, the limit is lower
// (tools/grad_aug_bench.py, us: n = 1438 603 vs 1128 alone, 1216 vs 1980 for four chains; 2048: 979 vs 1482, 2155 vs 2720;
// 2560: 1470 vs 1824, 3636 vs 3497; 3072: 2152 vs 2276, 5550 vs 4566; 3584: 2953 vs 2797)
static bool grad_augmented(const gpmi_ctx *c, int n, bool many)
{
    const int lim = many ? c->tune.grad_aug_ng : c->tune.grad_aug_n;
    return lim > 0 && n <= lim;
}

// every buffer logml_grad_core(c, ., n, ...) uses, at its final size
static int logml_grad_reserve(gpmi_ctx *c, int n, int D, bool many = false)
{
    const int ns = grad_ns(D);
    int rc;
    if (grad_augmented(c, n, many)) {
        if ((rc = reserve_ws(c, 2 * n + 1, 2 * n + 1))) return rc;
        const size_t T = (size_t)((n + 63) / 64), ntiles = T * (T + 1) / 2;
        double *b;
        return stage_buf(c, 3, ((size_t)GRAD_NS_MAX + ntiles * ns) * sizeof(double), &b);
    }
    if ((rc = reserve_ws(c, n + 1, n))) return rc;
    const size_t ldu = (size_t)(((n + 15) / 16) * 16 + 16);
    const int npan = (n + GPMI_NB - 1) / GPMI_NB;
    const size_t T = (size_t)((n + 63) / 64), ntiles = T * (T + 1) / 2;
    const int nchunk = (n + UMV_COLS - 1) / UMV_COLS;
    double *b;
    if ((rc = stage_buf(c, 2, ldu * (size_t)(n + 1) * sizeof(double), &b))) return rc;
    if ((rc = stage_buf(c, 3, (2 * (size_t)n + GRAD_NS_MAX + ntiles * ns + (size_t)nchunk * n) * sizeof(double), &b))) return rc;
    return scratch_buf(c, (size_t)npan * GPMI_FPACK * sizeof(double), &b);
}

// Device part of one value + gradient evaluation, enqueued on c->stream without synchronisation:
// d_res[0..2] = (logml, sum log L_ii, z'z), d_res[3 .. 3 + grad_ns(D)) = the contraction sums, *d_info = status
constexpr int GRAD_RES = 3 + GRAD_NS_MAX;
static int logml_grad_core(gpmi_ctx *c, const double *dX, int n, int ldx, const double *dy, const SeParams &p, double diag_add,
                           double *d_res, int *d_info, bool many = false)
{
    int rc;
    if ((rc = logml_grad_reserve(c, n, p.D, many))) return rc;
    const int ns = grad_ns(p.D);
    if (grad_augmented(c, n, many)) {
        const int M2 = 2 * n + 1;
        const size_t ld = (size_t)c->ld;
        const size_t T = (size_t)((n + 63) / 64), ntiles = T * (T + 1) / 2;
        double *sums = c->stage[3], *part = sums + GRAD_NS_MAX;
        hipStream_t s = c->stream;
        HIPCHK(hipMemsetAsync(c->d_info, 0, sizeof(int), s));
        launch_se_cov(c, s, dX, n, ldx, nullptr, n, ldx, p, diag_add, 1, c->W, ld);
        launch_set_row(s, c->W, ld, n, dy, n, n);
        hipLaunchKernelGGL(k_aug_identity, dim3((n + 63) / 64, (n + 15) / 16), 256, 0, s, c->W + n + 1, ld, n);
        HIPCHK(hipMemsetAsync(c->W + (size_t)n * ld, 0, ld * (size_t)(n + 1) * sizeof(double), s));   // columns n .. 2n
        if ((rc = launch_potrf_partial(c, c->W, ld, M2, M2, n, c->d_info, nullptr))) return rc;
        launch_logml_finalize(s, c->W, ld, n, n, c->d_info, d_res, d_info, c->d_fin);
        // trailing block at (n, n): [[-z'z, .], [-a, -K^-1]]; the products a_i a_j do not see a's sign
        const double *av = c->W + (size_t)(n + 1) + (size_t)n * ld;
        const double *Wk = c->W + (size_t)(n + 1) + (size_t)(n + 1) * ld;
        if (p.D <= GPMI_MAXD)
            hipLaunchKernelGGL(k_grad_partial, dim3((unsigned)T, (unsigned)T), 256, 0, s, dX, n, ldx, p, av, Wk, ld, part);
        else
            hipLaunchKernelGGL(k_grad_partial_big, dim3((unsigned)T, (unsigned)T), 256, 0, s, dX, n, ldx, p, av, Wk, ld, part, ns);
        hipLaunchKernelGGL(k_grad_final, dim3(1), 1024, 0, s, part, ntiles, d_res + 3, ns);
        HIPCHK(hipGetLastError());
        return 0;
    }
    const int M = n + 1;
    const size_t ld = (size_t)c->ld;
    const size_t ldu = (size_t)(((n + 15) / 16) * 16 + 16);
    const size_t T = (size_t)((n + 63) / 64), ntiles = T * (T + 1) / 2;
    const int nchunk = (n + UMV_COLS - 1) / UMV_COLS;
    double *U = c->stage[2], *vec = c->stage[3], *Fall = c->scratch;
    double *zv = vec, *av = vec + n, *sums = vec + 2 * (size_t)n, *part = sums + GRAD_NS_MAX, *mvpart = part + ntiles * ns;
    hipStream_t s = c->stream;
    // factorisation with the augmented row, all panel factors kept
    HIPCHK(hipMemsetAsync(c->d_info, 0, sizeof(int), s));
    launch_se_cov(c, s, dX, n, ldx, nullptr, n, ldx, p, diag_add, 1, c->W, ld);
    launch_set_row(s, c->W, ld, n, dy, n, n);
    if ((rc = launch_potrf_partial(c, c->W, ld, M, n, n, c->d_info, Fall))) return rc;
    launch_logml_finalize(s, c->W, ld, n, n, c->d_info, d_res, d_info, c->d_fin);
    launch_get_row(s, c->W, ld, n, 0, n, 1.0, zv);
    // U = L^-T, a = U z = K^-1 y
    hipLaunchKernelGGL(k_set_identity, dim3((n + 63) / 64, (n + 15) / 16), 256, 0, s, U, ldu, n);
    if ((rc = launch_trsm_right(c, c->W, ld, n, U, ldu, n, Fall, 1))) return rc;
    hipLaunchKernelGGL(k_upper_mv_part, dim3((n + UMV_ROWS - 1) / UMV_ROWS, nchunk), UMV_ROWS, 0, s, U, ldu, n, zv, mvpart);
    hipLaunchKernelGGL(k_upper_mv_sum, dim3((n + 255) / 256), 256, 0, s, mvpart, n, nchunk, av);
    // W(lower) = -U U^T = -K^-1
    HIPCHK(hipMemsetAsync(c->W, 0, ld * (size_t)n * sizeof(double), s));
    launch_syrk_uut(c, s, U, ldu, c->W, ld, n);
    if (p.D <= GPMI_MAXD)   // register-resident coordinates
        hipLaunchKernelGGL(k_grad_partial, dim3((unsigned)T, (unsigned)T), 256, 0, s, dX, n, ldx, p, av, c->W, ld, part);
    else                    // any D: LDS-staged coordinates, one pass for the distances and one per dimension
        hipLaunchKernelGGL(k_grad_partial_big, dim3((unsigned)T, (unsigned)T), 256, 0, s, dX, n, ldx, p, av, c->W, ld, part, ns);
    hipLaunchKernelGGL(k_grad_final, dim3(1), 1024, 0, s, part, ntiles, d_res + 3, ns);
    HIPCHK(hipGetLastError());
    return 0;
}

// host part: (d/dalpha, d/dell..., d/dsigma) from the contraction sums
static void logml_grad_finish(const double *hs, int D, double alpha, const double *ell, int n_ell, double sigma, double *grad)
{
    grad[0] = 2.0 * hs[0] / alpha;
    if (n_ell == 1) {
        double t = 0.0;
        for (int d = 0; d < D; ++d) t += hs[1 + d];
        grad[1] = t / (ell[0] * ell[0] * ell[0]);
    } else {
        for (int d = 0; d < D; ++d) grad[1 + d] = hs[1 + d] / (ell[d] * ell[d] * ell[d]);
    }
    grad[1 + n_ell] = 2.0 * sigma * hs[grad_ns(D) - 1];
}

extern "C" int gpmi_logml_grad(gpmi_ctx *c, const double *X, int n, int ldx, int D, const double *y,
                               double alpha, const double *ell, int n_ell, double sigma, double jitter,
                               double *out3, double *grad)
{
    ENTER(c);
    if (n <= 0 || !X || !y || !out3 || !grad || ldx < n || D < 1) return gpmi_fail(GPMI_EARG, "bad argument");
    SeParams p;
    int rc;
    if ((rc = fill_params(&p, D, alpha, ell, n_ell))) return rc;
    if (c->tune.small_ng1 > 0 && n <= c->tune.small_ng1 && D <= GPMI_MAXD) {
        // the sizes the reference's fits run at: ONE launch of one workgroup; X, y go in and the 13 results come back
        // through a pinned, device-mapped buffer (no copy call), as in gpmi_logml
        const size_t need = (16 + (size_t)n * (D + 1)) * sizeof(double);
        if ((rc = pin_reserve(c, need))) return rc;
        double *hX = c->h_pin + 16, *hy = hX + (size_t)n * D, *stage;
        for (int d = 0; d < D; ++d) memcpy(hX + (size_t)d * n, X + (size_t)d * ldx, (size_t)n * sizeof(double));
        memcpy(hy, y, (size_t)n * sizeof(double));
        if ((rc = scratch_buf(c, (size_t)n * (D + 1) * sizeof(double), &stage))) return rc;
        if ((rc = reserve_ws_small(c, n, 2))) return rc;
        double *pd = c->h_pin_dev;
        const int seq = ++c->pin_seq;
        launch_logml_grad_small(c->stream, pd + 16, n, n, pd + 16 + (size_t)n * D, p, sigma * sigma + jitter, c->W, pd,
                                (int *)(pd + GPMI_SMALL_GRAD_RES), c->d_info, stage, (int *)(pd + 15), seq);
        HIPCHK(hipGetLastError());
        if ((rc = pin_wait(c, (const int *)(c->h_pin + 15), seq))) return rc;
        const int info = *(const int *)(c->h_pin + GPMI_SMALL_GRAD_RES);
        for (int k = 0; k < 3; ++k) out3[k] = c->h_pin[k];
        if (info) {
            for (int k = 0; k < 2 + n_ell; ++k) grad[k] = NAN;
            return info;
        }
        logml_grad_finish(c->h_pin + 3, D, alpha, ell, n_ell, sigma, grad);
        return 0;
    }
    double *dX, *dy, *dres;
    if ((rc = upload_xy(c, X, n, ldx, D, y, &dX, &dy))) return rc;
    if ((rc = logml_grad_reserve(c, n, D))) return rc;
    dres = c->d_fin + 4096;  // second half of the finalize scratch (64 KiB): GRAD_RES doubles
    if ((rc = logml_grad_core(c, dX, n, n, dy, p, sigma * sigma + jitter, dres, c->d_info + 1))) return rc;
    double hr[GRAD_RES];
    int info = 0;
    hipStream_t s = c->stream;
    HIPCHK(hipMemcpyAsync(hr, dres, sizeof(hr), hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(&info, c->d_info + 1, sizeof(int), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    for (int k = 0; k < 3; ++k) out3[k] = hr[k];
    if (info) {
        for (int k = 0; k < 2 + n_ell; ++k) grad[k] = NAN;
        return info;
    }
    logml_grad_finish(hr + 3, D, alpha, ell, n_ell, sigma, grad);
    return 0;
}

// G independent (alpha[g], rho[g], sigma[g]) points: value AND gradient of each, concurrently on the lanes -- what
// rstan's default four chains ask for per leapfrog step (pendulum_fit.R:140: chains = 4, cores = 4), one
// evaluation + reverse sweep of models/fit_hyperparameters.stan:18-32 each.  out3: 3 G, grad: 3 G
// (d/dalpha, d/drho, d/dsigma per point), info: G (non-PD points: NaN and the grid continues).
extern "C" int gpmi_logml_grad_grid(gpmi_ctx *c, const double *X, int n, int ldx, int D, const double *y, const double *alpha,
                                    const double *rho, const double *sigma, int G, double jitter, double *out3, double *grad,
                                    int *info)
{
    ENTER(c);
    if (G < 0) return gpmi_fail(GPMI_EARG, "negative grid size");
    if (G == 0) return 0;
    if (n <= 0 || !X || !y || !alpha || !rho || !sigma || !out3 || !grad || !info || ldx < n || D < 1)
        return gpmi_fail(GPMI_EARG, "bad argument");
    int rc;
    std::vector<SeParams> ps(G);
    for (int g = 0; g < G; ++g)
        if ((rc = fill_params(&ps[g], D, alpha[g], &rho[g], 1))) return rc;
    if (D <= GPMI_MAXD && n <= (G >= 2 ? c->tune.small_ng : c->tune.small_ng1) && G <= 8) {
        // a sampler's handful of chains: ONE launch, X, y in and the results out through the pinned, device-mapped buffer
        // (no copy call, no stream synchronisation): [res (G x 13) | info (G ints) | flag | X | y]
        const size_t o_info = (size_t)G * GPMI_SMALL_GRAD_RES, o_flag = o_info + (G + 1) / 2, o_x = o_flag + 1;
        const size_t need = (o_x + (size_t)n * (D + 1)) * sizeof(double);
        if ((rc = pin_reserve(c, need))) return rc;
        double *hX = c->h_pin + o_x, *hy = hX + (size_t)n * D, *stage;
        for (int d = 0; d < D; ++d) memcpy(hX + (size_t)d * n, X + (size_t)d * ldx, (size_t)n * sizeof(double));
        memcpy(hy, y, (size_t)n * sizeof(double));
        if ((rc = scratch_buf(c, (size_t)G * n * (D + 1) * sizeof(double), &stage))) return rc;
        if ((rc = reserve_ws_small(c, n, 2 * G))) return rc;
        if ((rc = reserve_small_par(c, G))) return rc;
        double *pd = c->h_pin_dev;
        const int seq = ++c->pin_seq;
        launch_logml_grad_small_batch(c->stream, pd + o_x, n, n, D, pd + o_x + (size_t)n * D, alpha, rho, sigma, G, jitter, c->W, pd,
                                      (int *)(pd + o_info), c->d_sinfo, stage, (int *)(pd + o_flag), seq, c->d_ctr + 40);
        HIPCHK(hipGetLastError());
        if ((rc = pin_wait(c, (const int *)(c->h_pin + o_flag), seq))) return rc;
        const int *hinfo = (const int *)(c->h_pin + o_info);
        for (int g = 0; g < G; ++g) {
            const double *r = c->h_pin + (size_t)g * GPMI_SMALL_GRAD_RES;
            info[g] = hinfo[g];
            for (int k = 0; k < 3; ++k) out3[3 * g + k] = r[k];
            if (info[g]) {
                for (int k = 0; k < 3; ++k) grad[3 * g + k] = NAN;
            } else {
                logml_grad_finish(r + 3, D, alpha[g], &rho[g], 1, sigma[g], grad + 3 * g);
            }
        }
        return 0;
    }
    if (D <= GPMI_MAXD && n <= (G >= 2 ? c->tune.small_ng : c->tune.small_ng1)) {
        // one workgroup per point, up to GPMI_SMALL_PTS points per launch (parameters as kernel arguments)
        double *dX, *dy, *dres;
        if ((rc = upload_xy(c, X, n, ldx, D, y, &dX, &dy))) return rc;
        const int per = G < GPMI_SMALL_PTS ? G : GPMI_SMALL_PTS;
        if ((rc = reserve_ws_small(c, n, 2 * per))) return rc;
        if ((rc = reserve_small_par(c, G))) return rc;   // work ints (one per point of a launch)
        if ((rc = scratch_buf(c, ((size_t)G * (GPMI_SMALL_GRAD_RES + 1) + 8) * sizeof(double), &dres))) return rc;
        int *dinfo = (int *)(dres + (size_t)G * GPMI_SMALL_GRAD_RES);
        for (int g0 = 0; g0 < G; g0 += per) {
            const int gc = (G - g0 < per) ? G - g0 : per;
            launch_logml_grad_small_batch(c->stream, dX, n, n, D, dy, alpha + g0, rho + g0, sigma + g0, gc, jitter, c->W,
                                          dres + (size_t)g0 * GPMI_SMALL_GRAD_RES, dinfo + g0, c->d_sinfo);
        }
        std::vector<double> hr((size_t)G * GPMI_SMALL_GRAD_RES);
        HIPCHK(hipMemcpyAsync(hr.data(), dres, hr.size() * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipMemcpyAsync(info, dinfo, (size_t)G * sizeof(int), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        HIPCHK(hipGetLastError());
        for (int g = 0; g < G; ++g) {
            const double *r = hr.data() + (size_t)g * GPMI_SMALL_GRAD_RES;
            for (int k = 0; k < 3; ++k) out3[3 * g + k] = r[k];
            if (info[g]) {
                for (int k = 0; k < 3; ++k) grad[3 * g + k] = NAN;
            } else {
                logml_grad_finish(r + 3, D, alpha[g], &rho[g], 1, sigma[g], grad + 3 * g);
            }
        }
        return 0;
    }
    int lanes = c->grid_lanes > 0 ? c->grid_lanes : 4;
    if (lanes > 8) lanes = 8;
    if (lanes > G) lanes = G;
    if ((rc = lanes_prepare(c, lanes))) return rc;
    for (int k = 0; k < lanes; ++k)
        if ((rc = logml_grad_reserve(k ? c->lane[k - 1] : c, n, D, G > 1))) return rc;
    double *dX, *dy, *dres;
    if ((rc = upload_xy(c, X, n, ldx, D, y, &dX, &dy))) return rc;
    // the root context's scratch also holds its packed panel factors (logml_grad_reserve): the results live behind them
    const int npan = (n + GPMI_NB - 1) / GPMI_NB;
    if ((rc = scratch_buf(c, ((size_t)npan * GPMI_FPACK + (size_t)G * (GRAD_RES + 1) + 8) * sizeof(double), &dres))) return rc;
    dres += (size_t)npan * GPMI_FPACK;
    int *dinfo = (int *)(dres + (size_t)G * GRAD_RES);
    const int la_saved = c->lookahead;
    hipStream_t const caller = c->stream;
    lanes_fork(c, lanes, caller);
    for (int g = 0; g < G && !rc; ++g) {
        gpmi_ctx *lc = (g % lanes == 0) ? c : c->lane[g % lanes - 1];
        rc = logml_grad_core(lc, dX, n, n, dy, ps[g], sigma[g] * sigma[g] + jitter, dres + (size_t)g * GRAD_RES, dinfo + g, G > 1);
    }
    lanes_join(c, lanes, caller, la_saved);
    if (rc) return rc;
    std::vector<double> hr((size_t)G * GRAD_RES);
    HIPCHK(hipMemcpyAsync(hr.data(), dres, hr.size() * sizeof(double), hipMemcpyDeviceToHost, caller));
    HIPCHK(hipMemcpyAsync(info, dinfo, (size_t)G * sizeof(int), hipMemcpyDeviceToHost, caller));
    HIPCHK(hipStreamSynchronize(caller));
    HIPCHK(hipGetLastError());
    for (int g = 0; g < G; ++g) {
        const double *r = hr.data() + (size_t)g * GRAD_RES;
        for (int k = 0; k < 3; ++k) out3[3 * g + k] = r[k];
        if (info[g]) {
            for (int k = 0; k < 3; ++k) grad[3 * g + k] = NAN;
        } else {
            logml_grad_finish(r + 3, D, alpha[g], &rho[g], 1, sigma[g], grad + 3 * g);
        }
    }
    return 0;
}

// ---- GP posterior ---------------------------------------------------------------
extern "C" int gpmi_gp_condition(gpmi_ctx *c, const double *t, int n, const double *ts, int m,
                                 const double *y, double alpha, double l, double s2, double jitter,
                                 int kindK, int kindS, int kindSS, int flags, double *mn, double *Kn, int ldkn)
{
    ENTER(c);
    if (n <= 0 || m <= 0 || !t || !ts || !y || !mn || !Kn || ldkn < m || !(l > 0.0))
        return gpmi_fail(GPMI_EARG, "bad argument");
    if (kindK < 0 || kindK > GPMI_TT || kindS < 0 || kindS > GPMI_TT || kindSS < 0 || kindSS > GPMI_TT)
        return gpmi_fail(GPMI_EARG, "bad kernel kind");
    int rc;
    const int nt = n + m, M = nt + 1;
    if (c->tune.small_gc > 0 && M <= c->tune.small_gc) {
        // R/tests.R sizes: ONE launch of one workgroup; t, ts, y go in and mn, Kn, info come back through a pinned,
        // device-mapped buffer (no copy call): layout [info | mn (m) | Kn (m x m) | t | ts | y]
        const size_t need = (8 + (size_t)m + (size_t)m * m + 2 * (size_t)n + m) * sizeof(double);
        if ((rc = pin_reserve(c, need))) return rc;
        double *hmn = c->h_pin + 8, *hKn = hmn + m, *ht = hKn + (size_t)m * m, *hts = ht + n, *hy = hts + m, *stage;
        memcpy(ht, t, (size_t)n * sizeof(double));
        memcpy(hts, ts, (size_t)m * sizeof(double));
        memcpy(hy, y, (size_t)n * sizeof(double));
        if ((rc = scratch_buf(c, (2 * (size_t)n + m) * sizeof(double), &stage))) return rc;
        if ((rc = reserve_ws_small(c, nt, 1))) return rc;
        double *pd = c->h_pin_dev;
        const size_t o_mn = 8, o_Kn = o_mn + m, o_t = o_Kn + (size_t)m * m, o_ts = o_t + n, o_y = o_ts + m;
        launch_gp_condition_small(c->stream, pd + o_t, n, pd + o_ts, m, pd + o_y, kindK, kindS, kindSS, (flags & GPMI_COMPAT_RR) ? 1 : 0,
                                  alpha * alpha, l * l, s2, jitter, c->W, pd + o_Kn, (size_t)m, pd + o_mn, (int *)pd, c->d_info, stage,
                                  (int *)(pd + 7), ++c->pin_seq);
        HIPCHK(hipGetLastError());
        if ((rc = pin_wait(c, (const int *)(c->h_pin + 7), c->pin_seq))) return rc;
        memcpy(mn, hmn, (size_t)m * sizeof(double));
        for (int j = 0; j < m; ++j) memcpy(Kn + (size_t)j * ldkn, hKn + (size_t)j * m, (size_t)m * sizeof(double));
        return *(const int *)c->h_pin;
    }
    if ((rc = reserve_ws(c, M, nt))) return rc;
    const size_t ld = (size_t)c->ld;
    double *dt, *dts, *dy, *dKn;
    if ((rc = stage_buf(c, 0, (size_t)(n + m) * sizeof(double), &dt))) return rc;
    dts = dt + n;
    if ((rc = stage_buf(c, 1, (size_t)n * sizeof(double), &dy))) return rc;
    const int ldo = (m + 1) & ~1;
    if ((rc = stage_buf(c, 2, (size_t)ldo * (m + 1) * sizeof(double), &dKn))) return rc;
    hipStream_t s = c->stream;
    HIPCHK(hipMemcpyAsync(dt, t, (size_t)n * sizeof(double), hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(dts, ts, (size_t)m * sizeof(double), hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(dy, y, (size_t)n * sizeof(double), hipMemcpyHostToDevice, s));
    HIPCHK(hipMemsetAsync(c->d_info, 0, sizeof(int), s));
    const double a2 = alpha * alpha;
    const int compat = (flags & GPMI_COMPAT_RR) ? 1 : 0;
    // [[K + s2 I, .], [Ks, Kss]] lower-stored, then the row [y^T, 0]
    launch_deriv_cov(s, kindK, dt, n, dt, n, a2, l, compat, 1, c->W, ld);
    launch_add_diag(s, c->W, ld, n, s2);
    launch_deriv_cov(s, kindS, dts, m, dt, n, a2, l, compat, 0, c->W + n, ld);
    launch_deriv_cov(s, kindSS, dts, m, dts, m, a2, l, compat, 1, c->W + n + (size_t)n * ld, ld);
    launch_set_row(s, c->W, ld, nt, dy, n, nt);
    // factor the first n columns: rows n.. become Ks L^-T, the trailing block the Schur
    // complement Kss - Ks (K+s2 I)^-1 Ks^T, the last row [z^T, -mn^T]
    if ((rc = launch_potrf_partial(c, c->W, ld, M, nt, n, c->d_info, nullptr))) return rc;
    launch_copy_matrix(s, c->W + n + (size_t)n * ld, ld, dKn, (size_t)ldo, m, m, 2);
    launch_add_diag(s, dKn, (size_t)ldo, m, jitter);
    double *dmn = dKn + (size_t)ldo * m;  // spare column of the output staging
    launch_get_row(s, c->W, ld, nt, n, m, -1.0, dmn);
    HIPCHK(hipGetLastError());
    int info = 0;
    HIPCHK(hipMemcpyAsync(&info, c->d_info, sizeof(int), hipMemcpyDeviceToHost, s));
    if ((rc = d2h_matrix(c, dKn, (size_t)ldo, m, m, Kn, ldkn))) return rc;
    HIPCHK(hipMemcpyAsync(mn, dmn, (size_t)m * sizeof(double), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    return info;
}

// ---- sample_derivs: one draw of the derivative process, fused on the device --------------------
// pendulum_fit.R:227-255 (lorenz.Rmd:80-107 with separate prediction times): K = a^2 QQ(ti, ti),
// KsK = a^2 RQ(tis, ti), KsKs = a^2 RR(tis, tis); mu = KsK (K + sy^2 I)^-1 y (:242-245);
// cov = KsKs - KsK (K + sy^2 I)^-1 t(KsK) + jitter I (:247-251); one draw mvrnorm(1, mu, cov) (:253).
// Here: the augmented partial factorisation of gpmi_gp_condition leaves cov as the trailing block of the
// workspace; it is factored IN PLACE and the draw is mu + chol(cov) z -- the m x m covariance (537 MB at
// m = 8192) never crosses PCIe, where the host-side composition potrf(gp_condition(...)) moved it three
// times.  MASS::mvrnorm draws through an eigen-decomposition with R's (unseeded) RNG: the moments are the
// parity surface, the standard-normal variate z is the caller's.
namespace {
__global__ void k_axpy1(double *__restrict__ y, const double *__restrict__ x, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] += x[i];
}
// status of one draw from the two device infos: K + sy^2 I not PD -> its order k (1..n); cov not PD -> n + k
__global__ void k_merge_info(const int *__restrict__ a, const int *__restrict__ b, int n, int *__restrict__ out)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = a[0] ? a[0] : (b[0] ? n + b[0] : 0);
}
}  // namespace

// every buffer sample_derivs_core(c, n, m) touches, at its final size (a growing buffer synchronises its stream)
static int sample_derivs_reserve(gpmi_ctx *c, int n, int m)
{
    int rc;
    if ((rc = reserve_ws(c, n + m + 1, n + m))) return rc;
    double *b;
    return stage_buf(c, 3, ((size_t)trmv_lower_chunks(m) + 1) * m * sizeof(double), &b);
}

// all pointers device; ddraw, dmu: m doubles; d_status: 1 int; enqueues on c->stream, no synchronisation
static int sample_derivs_core(gpmi_ctx *c, const double *dt, int n, const double *dts, int m, const double *dy,
                              double a, double l, double s2, double jitter, const double *dz, double *ddraw, double *dmu,
                              int *d_status)
{
    int rc;
    const int nt = n + m, M = nt + 1;
    if ((rc = sample_derivs_reserve(c, n, m))) return rc;
    const size_t ld = (size_t)c->ld;
    hipStream_t s = c->stream;
    double *part = c->stage[3];
    HIPCHK(hipMemsetAsync(c->d_info, 0, 4 * sizeof(int), s));
    const double a2 = a * a;
    launch_deriv_cov(s, GPMI_QQ, dt, n, dt, n, a2, l, 0, 1, c->W, ld);
    launch_add_diag(s, c->W, ld, n, s2);
    launch_deriv_cov(s, GPMI_RQ, dts, m, dt, n, a2, l, 0, 0, c->W + n, ld);
    launch_deriv_cov(s, GPMI_RR, dts, m, dts, m, a2, l, 0, 1, c->W + n + (size_t)n * ld, ld);
    launch_set_row(s, c->W, ld, nt, dy, n, nt);
    if ((rc = launch_potrf_partial(c, c->W, ld, M, nt, n, c->d_info, nullptr))) return rc;
    double *S = c->W + n + (size_t)n * ld;     // Schur complement = cov - jitter I (lower), rows n .. nt - 1
    launch_add_diag(s, S, ld, m, jitter);
    launch_get_row(s, c->W, ld, nt, n, m, -1.0, dmu);  // last row of the trailing block: -mu^T
    if ((rc = launch_potrf_partial(c, S, ld, m, m, m, c->d_info + 2, nullptr))) return rc;
    launch_trmv_lower(s, S, ld, m, dz, ddraw, part);
    hipLaunchKernelGGL(k_axpy1, dim3((m + 255) / 256), 256, 0, s, ddraw, dmu, m);
    hipLaunchKernelGGL(k_merge_info, dim3(1), 64, 0, s, c->d_info, c->d_info + 2, n, d_status);
    HIPCHK(hipGetLastError());
    return 0;
}

// one workgroup per draw (k_sample_derivs_small_batch) instead of a launch chain per draw on the lanes: a workgroup needs
// ~(n + m)^3 time but every draw has a CU of its own (tools/sample_derivs_bench.py)
static bool small_sample_derivs(const gpmi_ctx *c, int n, int m, int B)
{
    const int M = n + m + 1;
    if (c->tune.small_sd <= 0 || M > c->tune.small_sd) return false;
    const double r = (double)M / 400.0;
    return (double)B >= (double)c->tune.small_sdb * r * r;
}

// device pointers, packed columns; enqueues on c->stream
static int sample_derivs_small(gpmi_ctx *c, const double *dt, int n, const double *dts, int m, const double *dY, const double *params,
                               int B, double jitter, const double *dZ, double *dD, double *dM, int *dst)
{
    int rc;
    const int per = B < GPMI_SMALL_DEV_PTS ? B : GPMI_SMALL_DEV_PTS;
    if ((rc = reserve_ws_small(c, n + m, per))) return rc;
    if ((rc = reserve_small_par(c, 2 * per))) return rc;
    for (int b0 = 0; b0 < B; b0 += per) {
        const int bc = (B - b0 < per) ? B - b0 : per;
        launch_sample_derivs_small_batch(c->stream, dt, n, dts, m, dY + (size_t)b0 * n, params + 3 * (size_t)b0, bc, jitter,
                                         dZ + (size_t)b0 * m, c->d_spar, c->W, dD + (size_t)b0 * m, dM + (size_t)b0 * m, dst + b0,
                                         c->d_sinfo);
    }
    HIPCHK(hipGetLastError());
    return 0;
}

extern "C" int gpmi_sample_derivs(gpmi_ctx *c, const double *t, int n, const double *ts, int m, const double *y,
                                  double l, double a, double sy, double jitter, const double *z, double *draw, double *mu)
{
    ENTER(c);
    if (n <= 0 || m <= 0 || !t || !ts || !y || !z || !draw || !(l > 0.0)) return gpmi_fail(GPMI_EARG, "bad argument");
    int rc;
    double *d;
    if ((rc = sample_derivs_reserve(c, n, m))) return rc;
    if ((rc = stage_buf(c, 0, ((size_t)2 * n + 4 * (size_t)m + 8) * sizeof(double), &d))) return rc;
    double *dt = d, *dy = d + n, *dts = dy + n, *dz = dts + m, *ddraw = dz + m, *dmu = ddraw + m;
    int *dst = (int *)(dmu + m);
    hipStream_t s = c->stream;
    HIPCHK(hipMemcpyAsync(dt, t, (size_t)n * sizeof(double), hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(dy, y, (size_t)n * sizeof(double), hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(dts, ts, (size_t)m * sizeof(double), hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(dz, z, (size_t)m * sizeof(double), hipMemcpyHostToDevice, s));
    if (small_sample_derivs(c, n, m, 1)) {   // small enough that one workgroup beats the launch chain even for ONE draw
        const double par[3] = {l, a, sy};
        if ((rc = sample_derivs_small(c, dt, n, dts, m, dy, par, 1, jitter, dz, ddraw, dmu, dst))) return rc;
    } else if ((rc = sample_derivs_core(c, dt, n, dts, m, dy, a, l, sy * sy, jitter, dz, ddraw, dmu, dst))) return rc;
    int st = 0;
    HIPCHK(hipMemcpyAsync(draw, ddraw, (size_t)m * sizeof(double), hipMemcpyDeviceToHost, s));
    if (mu) HIPCHK(hipMemcpyAsync(mu, dmu, (size_t)m * sizeof(double), hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(&st, dst, sizeof(int), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    return st;
}

// B independent draws, draw b from (params[3 b .. 3 b + 2] = (l, a, sy), Y[:, b], Z[:, b]): the loop
// mclapply(s_list[1:100], sample_derivs_both_states, mc.cores = 2) of pendulum_fit.R:261-268 -- one posterior
// draw of the hyper-parameters and one noisy series per call -- as concurrent conditionings on the lanes.
extern "C" int gpmi_sample_derivs_batch(gpmi_ctx *c, const double *t, int n, const double *ts, int m, const double *Y, int ldy,
                                        const double *params, int B, double jitter, const double *Z, int ldz, double *draws,
                                        int ldd, double *mus, int ldmu, int *info)
{
    ENTER(c);
    if (B < 0) return gpmi_fail(GPMI_EARG, "negative batch size");
    if (B == 0) return 0;
    if (n <= 0 || m <= 0 || !t || !ts || !Y || !params || !Z || !draws || !info || ldy < n || ldz < m || ldd < m || (mus && ldmu < m))
        return gpmi_fail(GPMI_EARG, "bad argument");
    for (int b = 0; b < B; ++b)
        if (!(params[3 * b] > 0.0)) return gpmi_fail(GPMI_EARG, "length-scale must be positive");
    int rc;
    const bool small = small_sample_derivs(c, n, m, B);
    int lanes = c->grid_lanes > 0 ? c->grid_lanes : 4;
    if (lanes > 8) lanes = 8;
    if (lanes > B) lanes = B;
    if (!small) {
        if ((rc = lanes_prepare(c, lanes))) return rc;
        for (int l = 0; l < lanes; ++l)
            if ((rc = sample_derivs_reserve(l ? c->lane[l - 1] : c, n, m))) return rc;
    }
    double *d;
    const size_t nb = (size_t)n * B, mb = (size_t)m * B;
    if ((rc = stage_buf(c, 0, ((size_t)n + m + nb + 3 * mb + B + 8) * sizeof(double), &d))) return rc;
    double *dt = d, *dts = dt + n, *dY = dts + m, *dZ = dY + nb, *dD = dZ + mb, *dM = dD + mb;
    int *dst = (int *)(dM + mb);
    hipStream_t const caller = c->stream;
    HIPCHK(hipMemcpyAsync(dt, t, (size_t)n * sizeof(double), hipMemcpyHostToDevice, caller));
    HIPCHK(hipMemcpyAsync(dts, ts, (size_t)m * sizeof(double), hipMemcpyHostToDevice, caller));
    HIPCHK(hipMemcpy2DAsync(dY, (size_t)n * sizeof(double), Y, (size_t)ldy * sizeof(double), (size_t)n * sizeof(double), B, hipMemcpyHostToDevice, caller));
    HIPCHK(hipMemcpy2DAsync(dZ, (size_t)m * sizeof(double), Z, (size_t)ldz * sizeof(double), (size_t)m * sizeof(double), B, hipMemcpyHostToDevice, caller));
    if (small) {
        if ((rc = sample_derivs_small(c, dt, n, dts, m, dY, params, B, jitter, dZ, dD, dM, dst))) return rc;
    } else {
        const int la_saved = c->lookahead;
        lanes_fork(c, lanes, caller);
        for (int b = 0; b < B && !rc; ++b) {
            gpmi_ctx *lc = (b % lanes == 0) ? c : c->lane[b % lanes - 1];
            const double l = params[3 * b], a = params[3 * b + 1], sy = params[3 * b + 2];
            rc = sample_derivs_core(lc, dt, n, dts, m, dY + (size_t)b * n, a, l, sy * sy, jitter, dZ + (size_t)b * m, dD + (size_t)b * m,
                                    dM + (size_t)b * m, dst + b);
        }
        lanes_join(c, lanes, caller, la_saved);
        if (rc) return rc;
    }
    HIPCHK(hipMemcpy2DAsync(draws, (size_t)ldd * sizeof(double), dD, (size_t)m * sizeof(double), (size_t)m * sizeof(double), B, hipMemcpyDeviceToHost, caller));
    if (mus) HIPCHK(hipMemcpy2DAsync(mus, (size_t)ldmu * sizeof(double), dM, (size_t)m * sizeof(double), (size_t)m * sizeof(double), B, hipMemcpyDeviceToHost, caller));
    HIPCHK(hipMemcpyAsync(info, dst, (size_t)B * sizeof(int), hipMemcpyDeviceToHost, caller));
    HIPCHK(hipStreamSynchronize(caller));
    HIPCHK(hipGetLastError());
    return 0;
}

// ---- sequential conditional sampler -------------------------------------
// create_p_dotXnS, R/ode_gp_library.R:43-93.  The reference re-derives, at every call, the joint
// mean and covariance of ALL star points so far,
//   m = K_XsX K~^-1 mn,   K = K_XsXs - K_XsX K~^-1 K_XXs + K_XsX K~^-1 Kn K~^-1 K_XXs   (:74-76)
// (K~ = K_XX + 1e-6 I through a QR factorisation, :55-57), symmetrises it, adds 1e-6 I (:77) and
// conditions the newest point on the draws already made with condMVN (:80-81): O(i N^2 + i^3) per
// call.  Here everything is expressed through the Cholesky factor K~ = L L^T and whitened kernel
// rows t_i = L^-1 k(X, xs_i) (norm <= alpha: no cancellation of 1/jitter-sized numbers, unlike an
// explicit K~^-1, which loses cond(K~) * eps * |K~^-1| -- 1e-5 in the R/tests.R:78 scenario):
//   once, on the device:  B = I - L^-1 Kn L^-T  (an N-row panel solve and one for the lower triangle alone),  b = L^-1 mn;
//   per call:  t_i (one one-row panel solve, reads the factor once), u = B t_i (HBM-bound
//   mat-vec), K[i,j] = k(xs_i, xs_j) - t_j . u,  m_i = t_i . b  (i + 2 dot products), and one row
//   appended to the Cholesky factor of the star covariance: with K[g,g] = Ls Ls^T and
//   l = Ls^-1 K[g,i],  condMean = m_i + l . w,  condVar = K[i,i] - l . l,  w = Ls^-1 (dots - m_g)
//   -- the same numbers as condMVN's solve, without re-factoring.
struct gpmi_seq {
    gpmi_ctx *c;
    int n, D, max_steps, i, pending;
    SeParams p;
    double jitter;
    size_t ldm;
    double *dX, *L, *Fall, *Dinv, *B, *a, *Kx, *Xs, *Ls, *u, *part, *ks, *w, *res, *kcol, *row2;
    int nchunk;
};

namespace {
constexpr int MV_ROWS = 256, MV_COLS = 512;

// B = I - G for a symmetric G of which only the LOWER triangle was computed: from G (lower valid) and its
// transposed copy Gt (upper valid), both read along columns; B is symmetric by construction
__global__ __launch_bounds__(256) void k_seq_b(const double *__restrict__ G, const double *__restrict__ Gt, size_t ld,
                                               double *__restrict__ Bm, int n)
{
    const int r = blockIdx.x * 64 + (threadIdx.x & 63);
    const int c0 = blockIdx.y * 16 + (threadIdx.x >> 6) * 4;
    if (r >= n) return;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int c = c0 + q;
        if (c < n) {
            const size_t o = (size_t)r + (size_t)c * ld;
            Bm[o] = (r == c ? 1.0 : 0.0) - (r >= c ? G[o] : Gt[o]);
        }
    }
}

// part[chunk][row] = sum over the chunk's columns of A[row, col] x[col]; thread = row (a wave
// reads 512 contiguous bytes of each column), four accumulators in a fixed order
__global__ __launch_bounds__(MV_ROWS) void k_mv_part(const double *__restrict__ A, size_t lda, int n,
                                                     const double *__restrict__ x, double *__restrict__ part)
{
    const int r = blockIdx.x * MV_ROWS + threadIdx.x;
    const int c0 = blockIdx.y * MV_COLS;
    const int c1 = (c0 + MV_COLS < n) ? c0 + MV_COLS : n;
    if (r >= n) return;
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    const double *col = A + (size_t)r + (size_t)c0 * lda;
    int c = c0;
    for (; c + 4 <= c1; c += 4) {
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[q] = fma(col[(size_t)q * lda], x[c + q], acc[q]);
        col += 4 * lda;
    }
    for (; c < c1; ++c) {
        acc[0] = fma(col[0], x[c], acc[0]);
        col += lda;
    }
    part[(size_t)blockIdx.y * n + r] = (acc[0] + acc[1]) + (acc[2] + acc[3]);
}

__global__ __launch_bounds__(256) void k_mv_sum(const double *__restrict__ part, int n, int nchunk, double scale,
                                                double *__restrict__ y)
{
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= n) return;
    double acc = 0.0;
    for (int q = 0; q < nchunk; ++q) acc += part[(size_t)q * n + r];
    y[r] = scale * acc;
}

// Kx holds the whitened rows t_j.  block j <= i: ks[j] = k(xs_i, xs_j) - t_j . u (+ jitter on
// j == i); block i + 1: ks[max_steps] = t_i . a  (the prior mean of the new point).  Fixed-shape tree: deterministic.
__global__ __launch_bounds__(256) void k_seq_dots(const double *__restrict__ Kx, int n, int i,
                                                  const double *__restrict__ u, const double *__restrict__ a,
                                                  const double *__restrict__ Xs, int max_steps, SeParams p,
                                                  double jitter, double *__restrict__ ks)
{
    __shared__ double red[256];
    const int j = blockIdx.x;
    const double *col = Kx + (size_t)(j <= i ? j : i) * n;
    const double *v = (j <= i) ? u : a;
    double acc = 0.0;
    for (int r = threadIdx.x; r < n; r += 256) acc = fma(col[r], v[r], acc);
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int h = 128; h > 0; h >>= 1) {
        if ((int)threadIdx.x < h) red[threadIdx.x] += red[threadIdx.x + h];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        if (j > i) {
            ks[max_steps] = red[0];
        } else {
            double e = 0.0;
            for (int d = 0; d < p.D; ++d) {
                const double r = (Xs[(size_t)i + (size_t)d * max_steps] - Xs[(size_t)j + (size_t)d * max_steps]) * p.inv_ell[d];
                e += r * r;
            }
            ks[j] = p.a2 * exp(-0.5 * e) - red[0] + (j == i ? jitter : 0.0);
        }
    }
}

// One workgroup: l = Ls^-1 ks[0:i) by column-oriented forward substitution in LDS, then
// res = (condMean, condVar, sqrt(condVar)), row i of Ls.  info: 1 + i when condVar <= 0.
__global__ __launch_bounds__(256) void k_seq_cond(double *__restrict__ Ls, int ldl, int i,
                                                  const double *__restrict__ ks, int max_steps,
                                                  const double *__restrict__ w, double *__restrict__ res,
                                                  int *__restrict__ info)
{
    extern __shared__ double b[];
    __shared__ double red[2][256];
    for (int t = threadIdx.x; t < i; t += 256) b[t] = ks[t];
    __syncthreads();
    for (int j = 0; j < i; ++j) {
        if (threadIdx.x == 0) b[j] /= Ls[(size_t)j + (size_t)j * ldl];
        __syncthreads();
        const double bj = b[j];
        for (int t = j + 1 + threadIdx.x; t < i; t += 256) b[t] = fma(-Ls[(size_t)t + (size_t)j * ldl], bj, b[t]);
        __syncthreads();
    }
    double ll = 0.0, lw = 0.0;
    for (int t = threadIdx.x; t < i; t += 256) {
        ll = fma(b[t], b[t], ll);
        lw = fma(b[t], w[t], lw);
        Ls[(size_t)i + (size_t)t * ldl] = b[t];
    }
    red[0][threadIdx.x] = ll;
    red[1][threadIdx.x] = lw;
    __syncthreads();
    for (int h = 128; h > 0; h >>= 1) {
        if ((int)threadIdx.x < h) {
            red[0][threadIdx.x] += red[0][threadIdx.x + h];
            red[1][threadIdx.x] += red[1][threadIdx.x + h];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const double var = ks[i] - red[0][0];
        const double sd = sqrt(var);
        res[0] = ks[max_steps] + red[1][0];
        res[1] = var;
        res[2] = sd;
        Ls[(size_t)i + (size_t)i * ldl] = sd;
        *info = (var > 0.0) ? 0 : i + 1;
    }
}

__global__ void k_seq_commit(double *__restrict__ w, int i, double dot, const double *__restrict__ res)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) w[i] = (dot - res[0]) / res[2];
}
}  // namespace

static void seq_mv(gpmi_seq *q, hipStream_t s, const double *A, size_t lda, const double *x, double scale, double *y)
{
    const int n = q->n;
    hipLaunchKernelGGL(k_mv_part, dim3((n + MV_ROWS - 1) / MV_ROWS, q->nchunk), MV_ROWS, 0, s, A, lda, n, x, q->part);
    hipLaunchKernelGGL(k_mv_sum, dim3((n + 255) / 256), 256, 0, s, q->part, n, q->nchunk, scale, y);
}

extern "C" int gpmi_seq_destroy(gpmi_seq *q)
{
    if (!q) return 0;
    gpmi_ctx *c = q->c;
    if (c && c->pid == (int)getpid()) {
        (void)hipSetDevice(c->device);
        (void)hipStreamSynchronize(c->stream);
        if (q->dX) (void)hipFree(q->dX);
        if (q->L) (void)hipFree(q->L);
        if (q->Fall) (void)hipFree(q->Fall);
        if (q->Dinv) (void)hipFree(q->Dinv);
        if (q->B) (void)hipFree(q->B);
        if (q->Kx) (void)hipFree(q->Kx);
        if (q->Ls) (void)hipFree(q->Ls);
        if (q->u) (void)hipFree(q->u);
    }
    free(q);
    return 0;
}

extern "C" int gpmi_seq_create(gpmi_ctx *c, gpmi_seq **out, const double *X, int n, int ldx, int D,
                               const double *mn, const double *Kn, int ldkn, double alpha, const double *ell,
                               int n_ell, double jitter, int max_steps)
{
    ENTER(c);
    if (!out) return gpmi_fail(GPMI_EARG, "out pointer is NULL");
    *out = nullptr;
    if (n <= 0 || !X || !mn || !Kn || ldx < n || ldkn < n) return gpmi_fail(GPMI_EARG, "bad argument");
    if (max_steps < 1 || max_steps > 2048) return gpmi_fail(GPMI_EARG, "max_steps must be in 1..2048");
    SeParams p;
    int rc;
    if ((rc = fill_params(&p, D, alpha, ell, n_ell))) return rc;
    gpmi_seq *q = (gpmi_seq *)calloc(1, sizeof(gpmi_seq));
    if (!q) return gpmi_fail(GPMI_ENOMEM, "host allocation failed");
    q->c = c;
    q->n = n;
    q->D = D;
    q->max_steps = max_steps;
    q->p = p;
    q->jitter = jitter;
    q->ldm = (size_t)(((n + 15) / 16) * 16 + 16);
    q->nchunk = (n + MV_COLS - 1) / MV_COLS;
    const size_t ldm = q->ldm, ms = (size_t)max_steps;
    const int npan = (n + GPMI_NB - 1) / GPMI_NB;
    const size_t nvec = 3 * (size_t)n + (size_t)q->nchunk * n + 2 * ms + 32 + 2 * ((size_t)n + 1) + 4096;
#define SEQ_ALLOC(ptr, count)                                                                      \
    if (hipMalloc((void **)&(ptr), (count) * sizeof(double)) != hipSuccess) {                      \
        gpmi_seq_destroy(q);                                                                       \
        return gpmi_fail(GPMI_ENOMEM, "cannot allocate %zu bytes for the sampler", (size_t)(count) * sizeof(double)); \
    }
    SEQ_ALLOC(q->dX, (size_t)n * D + ms * D)
    SEQ_ALLOC(q->L, ldm * (size_t)(n + 1) + 4096)
    SEQ_ALLOC(q->Fall, (size_t)npan * GPMI_FPACK)
    SEQ_ALLOC(q->Dinv, (size_t)npan * GPMI_NB * GPMI_NB)
    SEQ_ALLOC(q->B, ldm * (size_t)(n + 1) + 4096)
    SEQ_ALLOC(q->Kx, (size_t)n * ms)
    SEQ_ALLOC(q->Ls, ms * ms)
    SEQ_ALLOC(q->u, nvec)
#undef SEQ_ALLOC
    q->Xs = q->dX + (size_t)n * D;
    q->a = q->u + n;
    q->kcol = q->a + n;
    q->part = q->kcol + n;
    q->ks = q->part + (size_t)q->nchunk * n;
    q->w = q->ks + ms + 1;
    q->res = q->w + ms;
    q->row2 = q->res + 16;  // 1 x n row vector with ld = 2, followed by the tile over-read slack
#define SEQ_TRY(expr)            \
    if ((rc = (expr))) {         \
        gpmi_seq_destroy(q);     \
        return rc;               \
    }
    hipStream_t s = c->stream;
    SEQ_TRY(reserve_ws(c, n, n))
    const size_t ld = (size_t)c->ld;
    double *dKn, *U;
    SEQ_TRY(stage_buf(c, 0, ldm * (size_t)(n + 1) * sizeof(double), &dKn))
    SEQ_TRY(stage_buf(c, 2, ldm * (size_t)(n + 1) * sizeof(double), &U))
    auto fail_hip = [&](hipError_t e, const char *what) {
        gpmi_seq_destroy(q);
        return gpmi_fail(GPMI_EHIP, "%s failed: %s", what, hipGetErrorString(e));
    };
    hipError_t e;
    if ((e = hipMemcpy2DAsync(q->dX, (size_t)n * sizeof(double), X, (size_t)ldx * sizeof(double),
                              (size_t)n * sizeof(double), D, hipMemcpyHostToDevice, s)) != hipSuccess)
        return fail_hip(e, "upload of X");
    if ((e = hipMemcpy2DAsync(dKn, ldm * sizeof(double), Kn, (size_t)ldkn * sizeof(double), (size_t)n * sizeof(double),
                              n, hipMemcpyHostToDevice, s)) != hipSuccess)
        return fail_hip(e, "upload of Kn");
    if ((e = hipMemcpyAsync(q->u, mn, (size_t)n * sizeof(double), hipMemcpyHostToDevice, s)) != hipSuccess)
        return fail_hip(e, "upload of mn");
    if ((e = hipMemsetAsync(c->d_info, 0, sizeof(int), s)) != hipSuccess) return fail_hip(e, "memset");
    if ((e = hipMemsetAsync(q->row2, 0, (2 * ((size_t)n + 1) + 4096) * sizeof(double), s)) != hipSuccess)
        return fail_hip(e, "memset");
    // K~ = K_XX + jitter I (:50,55), L = chol(K~) kept with its packed block factors
    launch_se_cov(c, s, q->dX, n, n, nullptr, n, n, p, jitter, 1, c->W, ld);
    SEQ_TRY(launch_potrf_partial(c, c->W, ld, n, n, n, c->d_info, q->Fall))
    launch_copy_matrix(s, c->W, ld, q->L, ldm, n, n, 1);
    launch_diag_inverses(s, q->Fall, n, q->Dinv, U);  // for the one-launch solves below and in every step
    // G = L^-1 Kn L^-T: (Kn L^-T), transposed, times L^-T again -- G is symmetric (the reference symmetrises it,
    // :76-77), so the second solve computes its lower triangle only (a third of the work) and B = I - G mirrors it
    SEQ_TRY(launch_trsm_right(c, q->L, ldm, n, dKn, ldm, n, q->Fall))
    launch_transpose(s, dKn, ldm, U, ldm, n, n);
    SEQ_TRY(launch_trsm_right(c, q->L, ldm, n, U, ldm, n, q->Fall, 2))
    launch_transpose(s, U, ldm, dKn, ldm, n, n);
    hipLaunchKernelGGL(k_seq_b, dim3((n + 63) / 64, (n + 15) / 16), 256, 0, s, U, dKn, ldm, q->B, n);
    // b = L^-1 mn (:56)
    SEQ_TRY(launch_trsv_lower(s, q->L, ldm, n, q->u, q->a, q->Dinv, c->d_ctr + 32))
    if ((e = hipGetLastError()) != hipSuccess) return fail_hip(e, "sampler set-up launch");
    int info = 0;
    if ((e = hipMemcpyAsync(&info, c->d_info, sizeof(int), hipMemcpyDeviceToHost, s)) != hipSuccess)
        return fail_hip(e, "download");
    if ((e = hipStreamSynchronize(s)) != hipSuccess) return fail_hip(e, "synchronise");
#undef SEQ_TRY
    if (info) {
        gpmi_seq_destroy(q);
        return info;
    }
    *out = q;
    return 0;
}

extern "C" int gpmi_seq_step(gpmi_seq *q, const double *xs, double *out2)
{
    if (!q) return gpmi_fail(GPMI_EARG, "sampler is NULL");
    gpmi_ctx *c = q->c;
    ENTER(c);
    if (!xs || !out2) return gpmi_fail(GPMI_EARG, "bad argument");
    if (q->i >= q->max_steps) return gpmi_fail(GPMI_EARG, "sampler is full (max_steps = %d)", q->max_steps);
    const int n = q->n, i = q->i, ms = q->max_steps;
    hipStream_t s = c->stream;
    // row i of Xs (leading dimension max_steps)
    HIPCHK(hipMemcpy2DAsync(q->Xs + i, (size_t)ms * sizeof(double), xs, sizeof(double), sizeof(double), q->D,
                            hipMemcpyHostToDevice, s));
    double *ti = q->Kx + (size_t)i * n;
    int rc;
    launch_se_cov(c, s, q->dX, n, n, q->Xs + i, 1, ms, q->p, 0.0, 0, q->kcol, (size_t)n);  // K_XsX row (:71)
    if ((rc = launch_trsv_lower(s, q->L, q->ldm, n, q->kcol, ti, q->Dinv, q->c->d_ctr + 32))) return rc;    // t_i = L^-1 k_i, one launch
    seq_mv(q, s, q->B, q->ldm, ti, 1.0, q->u);
    hipLaunchKernelGGL(k_seq_dots, dim3(i + 2), 256, 0, s, q->Kx, n, i, q->u, q->a, q->Xs, ms, q->p, q->jitter, q->ks);
    hipLaunchKernelGGL(k_seq_cond, dim3(1), 256, (size_t)(i + 1) * sizeof(double), s, q->Ls, ms, i, q->ks, ms, q->w,
                       q->res, c->d_info);
    HIPCHK(hipGetLastError());
    double res[3];
    int info = 0;
    HIPCHK(hipMemcpyAsync(res, q->res, sizeof(res), hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(&info, c->d_info, sizeof(int), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    out2[0] = res[0];
    out2[1] = res[1];
    q->pending = (info == 0);
    return info;
}

extern "C" int gpmi_seq_commit(gpmi_seq *q, double dot_xs)
{
    if (!q) return gpmi_fail(GPMI_EARG, "sampler is NULL");
    gpmi_ctx *c = q->c;
    ENTER(c);
    if (!q->pending) return gpmi_fail(GPMI_EARG, "gpmi_seq_commit without a preceding successful gpmi_seq_step");
    hipLaunchKernelGGL(k_seq_commit, dim3(1), 64, 0, c->stream, q->w, q->i, dot_xs, q->res);
    HIPCHK(hipGetLastError());
    q->i += 1;
    q->pending = 0;
    return 0;
}

extern "C" int gpmi_seq_count(const gpmi_seq *q) { return q ? q->i : 0; }

// ---- diagnostics ---------------------------------------------------------------
extern "C" int gpmi_last_timing(gpmi_ctx *c, double *ms3)
{
    ENTER(c);
    if (!ms3) return gpmi_fail(GPMI_EARG, "ms3 is NULL");
    for (int i = 0; i < 3; ++i) ms3[i] = c->last_ms[i];
    return 0;
}

// out[3*cat + 0..2] = launches, total ms, total work (bytes for cat 0, flops for 1 and 2) since
// the last reset; with n_out = 12, out[9..11] = the same for the trailing-update launches of more than one round
// of tiles (the throughput-bound subset of category 1; the single-round ones carry the next diagonal block's
// latency chain).  Synchronises the stream.
static int kernel_timing(gpmi_ctx *c, int reset, double *out, int n_out)
{
    if (!c->ktimer) return gpmi_fail(GPMI_EARG, "kernel timing was never enabled");
    KTimer *k = (KTimer *)c->ktimer;
    HIPCHK(hipStreamSynchronize(c->stream));
    if (out) {
        double sub[3] = {0.0, 0.0, 0.0};
        for (int cat = 0; cat < 3; ++cat) {
            double tot = 0.0;
            for (size_t i = 0; i < k->used[cat]; ++i) {
                float ms = 0.f;
                HIPCHK(hipEventElapsedTime(&ms, k->a[cat][i], k->b[cat][i]));
                tot += ms;
                if (cat == 1 && k->flag[cat][i]) {
                    sub[0] += 1.0;
                    sub[1] += ms;
                    sub[2] += k->w[cat][i];
                }
            }
            out[3 * cat] = (double)k->used[cat];
            out[3 * cat + 1] = tot;
            out[3 * cat + 2] = k->work[cat];
        }
        if (n_out >= 12)
            for (int i = 0; i < 3; ++i) out[9 + i] = sub[i];
    }
    if (reset)
        for (int cat = 0; cat < 3; ++cat) {
            k->used[cat] = 0;
            k->work[cat] = 0.0;
        }
    return 0;
}

extern "C" int gpmi_kernel_timing(gpmi_ctx *c, int reset, double *out9)
{
    ENTER(c);
    return kernel_timing(c, reset, out9, 9);
}

extern "C" int gpmi_kernel_timing_ex(gpmi_ctx *c, int reset, double *out12)
{
    ENTER(c);
    return kernel_timing(c, reset, out12, 12);
}

#ifdef GPMI_PROBES
// Stand-alone trailing-update launch on synthetic data: C (m x m, lower) -= P P^T, P m x k.
// `reps` back-to-back launches timed with HIP events; ms = average per launch.
namespace {
__global__ void k_fill(double *p, size_t n, double scale)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned long long x = i * 0x9E3779B97F4A7C15ULL + 0x1234567ULL;
        x ^= x >> 31; x *= 0xBF58476D1CE4E5B9ULL; x ^= x >> 29;
        p[i] = scale * ((double)(x >> 11) * (1.0 / 9007199254740992.0) - 0.5);
    }
}
}  // namespace
extern "C" int gpmi_probe_syrk(gpmi_ctx *c, int m, int k, int reps, double *ms)
{
    ENTER(c);
    if (m <= 0 || k <= 0 || reps <= 0 || !ms) return gpmi_fail(GPMI_EARG, "bad argument");
    int rc;
    if ((rc = reserve_ws(c, m, m))) return rc;
    const size_t ld = (size_t)c->ld;
    double *P;
    if ((rc = stage_buf(c, 3, ld * (size_t)(k + 1) * sizeof(double), &P))) return rc;
    hipLaunchKernelGGL(k_fill, dim3(2048), 256, 0, c->stream, c->W, ld * (size_t)m, 1.0);
    hipLaunchKernelGGL(k_fill, dim3(2048), 256, 0, c->stream, P, ld * (size_t)k, 1e-3);
    launch_syrk_probe(c, c->stream, P, ld, c->W, ld, m, k);
    HIPCHK(hipEventRecord(c->ev[0], c->stream));
    for (int r = 0; r < reps; ++r) launch_syrk_probe(c, c->stream, P, ld, c->W, ld, m, k);
    HIPCHK(hipEventRecord(c->ev[1], c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    HIPCHK(hipGetLastError());
    float t = 0.f;
    HIPCHK(hipEventElapsedTime(&t, c->ev[0], c->ev[1]));
    *ms = t / reps;
    return 0;
}

int probe_fused_read(hipStream_t s, unsigned long long *out5);
// out5: block 0 of the fused in-block launches since the last call: cycles in its 64 x 64 sub-tile, in the
// wait for the other two sub-tiles, in the diagonal-block body; launches; sum of K
extern "C" int gpmi_probe_fused(gpmi_ctx *c, double *out5)
{
    ENTER(c);
    unsigned long long h[8];
    if (probe_fused_read(c->stream, h)) return gpmi_fail(GPMI_EHIP, "probe read failed");
    for (int i = 0; i < 8; ++i) out5[i] = (double)h[i];   // [5..7]: body cycles, sub-tile cycles, launches of the K = 128 leaf launches
    return 0;
}

int probe_body_read(hipStream_t s, unsigned long long *out8);
// out6: diagonal-block bodies since the last call: cycles in the block loads, the 8-step loop, the stores (tile wave 0),
// inside factor16 and waiting for the next diagonal tile (factor wave); number of bodies
extern "C" int gpmi_probe_body(gpmi_ctx *c, double *out6)
{
    ENTER(c);
    unsigned long long h[8];
    if (probe_body_read(c->stream, h)) return gpmi_fail(GPMI_EHIP, "probe read failed");
    for (int i = 0; i < 6; ++i) out6[i] = (double)h[i];
    return 0;
}

// out6: phase cycles of the one-workgroup small-N kernels since the last call (block 0 of every launch): build,
// diagonal blocks, rows below, launches, trailing tiles, finalize
extern "C" int gpmi_probe_small(gpmi_ctx *c, double *out6)
{
    ENTER(c);
    unsigned long long h[8];
    if (probe_fused_read(c->stream, h)) return gpmi_fail(GPMI_EHIP, "probe read failed");
    for (int i = 0; i < 6; ++i) out6[i] = (double)h[i];
    return 0;
}

int probe_clock_read(hipStream_t s, int reset, unsigned long long *out3);
// out3: summed shader cycles, summed 100 MHz ticks, workgroups of all SYRK launches since the last reset
extern "C" int gpmi_probe_clock(gpmi_ctx *c, int reset, double *out3)
{
    ENTER(c);
    unsigned long long h[3];
    if (probe_clock_read(c->stream, reset, h)) return gpmi_fail(GPMI_EHIP, "clock probe read failed");
    if (out3)
        for (int i = 0; i < 3; ++i) out3[i] = (double)h[i];
    return 0;
}

extern "C" int gpmi_probe_mfma(gpmi_ctx *c, const double *A64, const double *B64, double *out256)
{
    ENTER(c);
    if (!A64 || !B64 || !out256) return gpmi_fail(GPMI_EARG, "bad argument");
    double *d;
    int rc;
    if ((rc = stage_buf(c, 0, 512 * sizeof(double), &d))) return rc;
    HIPCHK(hipMemcpyAsync(d, A64, 64 * sizeof(double), hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(d + 64, B64, 64 * sizeof(double), hipMemcpyHostToDevice, c->stream));
    launch_probe_mfma(c->stream, d, d + 64, d + 128);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(out256, d + 128, 256 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return 0;
}

extern "C" int gpmi_probe_mfma_peak(gpmi_ctx *c, int iters, double *tflops, double *clock_mhz)
{
    ENTER(c);
    if (iters <= 0 || !tflops) return gpmi_fail(GPMI_EARG, "bad argument");
    double *d;
    int rc, blocks = 0, threads = 0;
    if ((rc = stage_buf(c, 0, (8 + 2 * 1024) * sizeof(double), &d))) return rc;
    launch_probe_peak(c->stream, d, 16, &blocks, &threads);  // warm-up
    HIPCHK(hipEventRecord(c->ev[0], c->stream));
    launch_probe_peak(c->stream, d, iters, &blocks, &threads);
    HIPCHK(hipEventRecord(c->ev[1], c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, c->ev[0], c->ev[1]));
    const double flops = (double)blocks * (threads / 64) * (double)iters * 8.0 * 2048.0;
    *tflops = flops / (ms * 1e-3) / 1e12;
    if (clock_mhz) {
        std::vector<double> h(8 + 2 * (size_t)blocks);
        HIPCHK(hipMemcpy(h.data(), d, h.size() * sizeof(double), hipMemcpyDeviceToHost));
        double cyc = 0.0, real = 0.0;
        for (int b = 0; b < blocks; ++b) {
            cyc += h[8 + 2 * b];
            real += h[9 + 2 * b];
        }
        *clock_mhz = real > 0.0 ? cyc / real * 100.0 : 0.0;
    }
    return 0;
}
#endif  // GPMI_PROBES
