/*
 * Plain-C client of the libgpmi C ABI (include/gpmi.h): what an R .Call() shim, a cgo or a JNI
 * binding does, with no Python and no torch in the process.  Built and run by
 * tests/test_gpu_abi_client.py on the GPU box:
 *     gcc -O2 -Iinclude tests/abi_client.c -Lgp_amd/csrc -lgpmi -lm -Wl,-rpath,$PWD/gp_amd/csrc
 * Checks the known answers K1 (the R/tests.R:5 grid t = -2, -1.8, ..., 2 with y = exp(t);
 * tests/golden/kat.json, confirmed with mpmath at 50 digits) and K2 (c1-shaped N = 256) through
 * gpmi_logml, the non-positive-definite status code, and the QQ / RR kernels of R/kernels.R.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "gpmi.h"

static int fails = 0;
#define CHECK(cond, ...)                                  \
    do {                                                  \
        if (!(cond)) {                                    \
            fails++;                                      \
            printf("FAIL %s:%d: ", __FILE__, __LINE__);   \
            printf(__VA_ARGS__);                          \
            printf("\n");                                 \
        }                                                 \
    } while (0)

int main(void)
{
    gpmi_ctx *ctx = NULL;
    int rc = gpmi_create(&ctx, 0);
    if (rc) {
        printf("gpmi_create failed (%d): %s\n", rc, gpmi_last_error());
        return 2;
    }
    CHECK(gpmi_version() == GPMI_VERSION, "version %d", gpmi_version());

    /* K1 */
    double t[21], y[21], out[3], ell = 1.0;
    for (int i = 0; i < 21; ++i) {
        t[i] = -2.0 + 0.2 * i;
        y[i] = exp(t[i]);
    }
    rc = gpmi_logml(ctx, t, 21, 21, 1, y, 1.0, &ell, 1, 0.05, 0.0, out);
    CHECK(rc == 0, "K1 status %d (%s)", rc, gpmi_last_error());
    CHECK(fabs(out[0] - (-30.381144819459923)) <= 1e-8 * 30.381144819459923, "K1 logml %.15g", out[0]);
    CHECK(fabs(out[1] - (-43.52586771086652)) <= 1e-8 * 43.52586771086652, "K1 sum log diag %.15g", out[1]);

    /* K2: x = linspace(0, 10, 256), y = sin(x), alpha = rho = 1, sigma = 0.1 */
    enum { N2 = 256 };
    double x2[N2], y2[N2];
    for (int i = 0; i < N2; ++i) {
        x2[i] = 10.0 * i / (N2 - 1);
        y2[i] = sin(x2[i]);
    }
    rc = gpmi_logml(ctx, x2, N2, N2, 1, y2, 1.0, &ell, 1, 0.1, 0.0, out);
    CHECK(rc == 0, "K2 status %d", rc);
    CHECK(fabs(out[0] - 308.6859849221406) <= 1e-8 * 308.6859849221406, "K2 logml %.15g", out[0]);

    /* a matrix that is not positive definite: LAPACK-style status k = order of the failing minor */
    double A[9] = {4, 2, 0, 2, 1, 0, 0, 0, 1}; /* rank-deficient leading 2 x 2 block */
    rc = gpmi_potrf(ctx, A, 3, 3);
    CHECK(rc == 2, "potrf status %d, expected 2", rc);

    /* QQ and RR of R/kernels.R:22-32 at one pair, against the closed forms */
    double xa = 0.3, xb = 1.1, K = 0.0, l = 0.7, a = 1.3;
    rc = gpmi_deriv_cov(ctx, GPMI_QQ, &xa, 1, &xb, 1, a, l, GPMI_FULL, &K, 1);
    const double r = xa - xb, e = exp(-r * r / (2 * l * l));
    CHECK(rc == 0 && fabs(K - a * a * e) <= 1e-14, "QQ %.17g", K);
    rc = gpmi_deriv_cov(ctx, GPMI_RR, &xa, 1, &xb, 1, a, l, GPMI_FULL, &K, 1);
    CHECK(rc == 0 && fabs(K - a * a * e * (1 / (l * l) - r * r / (l * l * l * l))) <= 1e-13, "RR %.17g", K);

    /* bad argument: status, message, context still usable */
    rc = gpmi_logml(ctx, t, 21, 21, 1, y, 1.0, &ell, 3, 0.05, 0.0, out);
    CHECK(rc == GPMI_EARG && gpmi_last_error()[0] != 0, "bad-argument status %d", rc);
    rc = gpmi_logml(ctx, t, 21, 21, 1, y, 1.0, &ell, 1, 0.05, 0.0, out);
    CHECK(rc == 0, "context unusable after an argument error");

    gpmi_destroy(ctx);
    printf(fails ? "abi_client: %d check(s) failed\n" : "abi_client: ok\n", fails);
    return fails ? 1 : 0;
}
