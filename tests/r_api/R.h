/* tests/r_api/R.h -- NOT R's header.  Declarations of the part of R's public C API that r/gpmi_shim.c uses,
 * written from the API's documentation ("Writing R Extensions"), so that `gcc -fsyntax-only` can parse and
 * type-check the shim in an image without R (tests/test_abi.py).  It pins nothing about R's behaviour. */
#ifndef GPMI_TEST_R_H
#define GPMI_TEST_R_H
#include <stddef.h>
void Rf_error(const char *fmt, ...) __attribute__((noreturn, format(printf, 1, 2)));
char *R_alloc(size_t n, int size);
#endif
