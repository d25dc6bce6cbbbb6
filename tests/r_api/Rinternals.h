/* tests/r_api/Rinternals.h -- NOT R's header: see R.h next to it (syntax / type check of r/gpmi_shim.c only). */
#ifndef GPMI_TEST_RINTERNALS_H
#define GPMI_TEST_RINTERNALS_H
#include <stddef.h>
typedef struct SEXPREC *SEXP;
typedef ptrdiff_t R_xlen_t;
typedef unsigned int SEXPTYPE;
typedef enum { FALSE = 0, TRUE } Rboolean;
#define INTSXP 13
#define REALSXP 14
#define STRSXP 16
#define VECSXP 19
extern SEXP R_NilValue, R_NamesSymbol;
int TYPEOF(SEXP x);
double *REAL(SEXP x);
int *INTEGER(SEXP x);
int Rf_length(SEXP x);
R_xlen_t Rf_xlength(SEXP x);
int Rf_nrows(SEXP x);
int Rf_ncols(SEXP x);
double Rf_asReal(SEXP x);
int Rf_asInteger(SEXP x);
Rboolean Rf_isNull(SEXP x);
SEXP Rf_allocVector(SEXPTYPE type, R_xlen_t len);
SEXP Rf_allocMatrix(SEXPTYPE type, int nrow, int ncol);
SEXP Rf_duplicate(SEXP x);
SEXP Rf_protect(SEXP x);
void Rf_unprotect(int n);
#define PROTECT(x) Rf_protect(x)
#define UNPROTECT(n) Rf_unprotect(n)
SEXP VECTOR_ELT(SEXP x, R_xlen_t i);
SEXP SET_VECTOR_ELT(SEXP x, R_xlen_t i, SEXP v);
void SET_STRING_ELT(SEXP x, R_xlen_t i, SEXP v);
SEXP Rf_mkChar(const char *s);
SEXP Rf_setAttrib(SEXP x, SEXP name, SEXP val);
typedef void (*R_CFinalizer_t)(SEXP);
SEXP R_MakeExternalPtr(void *p, SEXP tag, SEXP prot);
void *R_ExternalPtrAddr(SEXP s);
void R_ClearExternalPtr(SEXP s);
void R_RegisterCFinalizerEx(SEXP s, R_CFinalizer_t fun, Rboolean onexit);
#endif
