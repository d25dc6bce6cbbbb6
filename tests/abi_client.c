/*
 * Plain-C client of the libgpmi C ABI (include/gpmi.h): what an R .Call() shim, a cgo or a JNI
 * binding does, with no Python and no torch in the process.  Built and run by
 * tests/test_gpu_abi_client.py on the GPU box:
 *     gcc -O2 -Iinclude tests/abi_client.c -Lgp_amd/csrc -lgpmi -lm -Wl,-rpath,$PWD/gp_amd/csrc
 * Checks the known answers K1 (the R/tests.R:5 grid t = -2, -1.8, ..., 2 with y = exp(t);
 * tests/golden/kat.json, confirmed with mpmath at 50 digits) and K2 (c1-shaped N = 256) through
 * gpmi_logml, the non-positive-definite status code, the QQ / RR kernels of R/kernels.R, value + gradient, a grid, and the
 * latent exact GP's transform.
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

    /* value + gradient (one launch of one workgroup at this size) against central differences of the value */
    {
        double g[3], v3[3], vp[3], vm[3];
        const double al = 1.1, rho = 0.9, sg = 0.07, h = 1e-6;
        double rr = rho;
        rc = gpmi_logml_grad(ctx, t, 21, 21, 1, y, al, &rr, 1, sg, 0.0, v3, g);
        CHECK(rc == 0, "logml_grad status %d (%s)", rc, gpmi_last_error());
        rr = rho + h; gpmi_logml(ctx, t, 21, 21, 1, y, al, &rr, 1, sg, 0.0, vp);
        rr = rho - h; gpmi_logml(ctx, t, 21, 21, 1, y, al, &rr, 1, sg, 0.0, vm);
        const double fd = (vp[0] - vm[0]) / (2 * h);
        CHECK(fabs(g[1] - fd) <= 1e-5 * (1.0 + fabs(fd)), "d logml / d rho %.10g vs central difference %.10g", g[1], fd);
        rr = rho;
        gpmi_logml(ctx, t, 21, 21, 1, y, al, &rr, 1, sg, 0.0, vp);
        CHECK(v3[0] == vp[0], "value of logml_grad %.17g != logml %.17g", v3[0], vp[0]);
        /* a grid of three points equals three single evaluations bit for bit */
        double ga[3] = {1.0, 1.1, 0.9}, gr[3] = {0.8, 0.9, 1.2}, gs[3] = {0.05, 0.07, 0.1}, go[9];
        int gi[3];
        rc = gpmi_logml_grid(ctx, t, 21, 21, 1, y, ga, gr, gs, 3, 0.0, go, gi);
        CHECK(rc == 0 && gi[0] == 0 && gi[1] == 0 && gi[2] == 0, "grid status %d", rc);
        for (int k = 0; k < 3; ++k) {
            gpmi_logml(ctx, t, 21, 21, 1, y, ga[k], &gr[k], 1, gs[k], 0.0, vp);
            CHECK(go[3 * k] == vp[0], "grid point %d: %.17g vs %.17g", k, go[3 * k], vp[0]);
        }
    }

    /* the latent exact GP's transform (models/exact_gp.stan:17-25): n = 1 is sqrt(alpha^2 + jitter) z; n = 21 against
     * f = L z composed from the separate entry points */
    {
        double x1 = 0.5, z1 = -2.0, f1 = 0.0, one = 1.0;
        rc = gpmi_exact_gp_f(ctx, &x1, 1, 1, 1, 1.5, &one, 1, 0.25, &z1, &f1);
        CHECK(rc == 0 && fabs(f1 - sqrt(1.5 * 1.5 + 0.25) * z1) <= 1e-15 * 4, "exact_gp_f n = 1: %.17g", f1);
        double Kc[21 * 21], zz[21], ff[21], fc[21];
        for (int i = 0; i < 21; ++i) zz[i] = sin(1.0 + i);
        rc = gpmi_exact_gp_f(ctx, t, 21, 21, 1, 1.0, &one, 1, 1e-6, zz, ff);
        CHECK(rc == 0, "exact_gp_f status %d", rc);
        rc = gpmi_se_cov(ctx, t, 21, 21, NULL, 0, 0, 1, 1.0, &one, 1, 1e-6, GPMI_FULL, Kc, 21);
        CHECK(rc == 0, "se_cov status %d", rc);
        rc = gpmi_potrf(ctx, Kc, 21, 21);
        CHECK(rc == 0, "potrf status %d", rc);
        rc = gpmi_trmv_lower(ctx, Kc, 21, 21, zz, fc);
        double worst = 0.0;
        for (int i = 0; i < 21; ++i) worst = fmax(worst, fabs(ff[i] - fc[i]));
        CHECK(rc == 0 && worst <= 1e-9, "exact_gp_f vs potrf + trmv: %.3g", worst);
    }

    /* bad argument: status, message, context still usable */
    rc = gpmi_logml(ctx, t, 21, 21, 1, y, 1.0, &ell, 3, 0.05, 0.0, out);
    CHECK(rc == GPMI_EARG && gpmi_last_error()[0] != 0, "bad-argument status %d", rc);
    rc = gpmi_logml(ctx, t, 21, 21, 1, y, 1.0, &ell, 1, 0.05, 0.0, out);
    CHECK(rc == 0, "context unusable after an argument error");

    gpmi_destroy(ctx);
    printf(fails ? "abi_client: %d check(s) failed\n" : "abi_client: ok\n", fails);
    return fails ? 1 : 0;
}
