/* bfref.c -- ORACLE (test infrastructure, not product code); see bfref.h.
 *
 * CPU restatement of the reference's apply path on ABI-compatible objects.
 * Each Mul below follows the reference function named in its comment:
 * the same loop order, the same view / multiply / accumulate steps and the
 * same heap traffic (one view struct per row range, one freshly allocated
 * result per leaf product), because the reference's CPU cost is dominated by
 * exactly those (SURVEY.md section 3.1 "hot loops").
 *
 * PARITY STATUS: parity unpinned by reference goldens (see bfref.h header).
 */
#define _GNU_SOURCE
#include "bfref.h"
#include "../include/bfhip.h"
#include "../include/bfhip_synth.h"

#include <complex.h>
#include <dlfcn.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef double _Complex cplx;

/* ---- error state (src/error.c:7-24, made non-fatal) ----------------------- */
static int currentError = 0;
int bfGetError(void) { return currentError; }
void bfClearError(void) { currentError = 0; }
static void setError(int e) { if (!currentError) currentError = e; }
/* src/error.c:20-24 under its own name (the assertion is dropped with the rest of the fatal
 * behaviour): what a foreign operator behind the vtable calls to raise the reference's error state */
void bfSetError(int error) { setError(error); }

static BfrefCounters counters;
void bfrefResetCounters(void) { memset(&counters, 0, sizeof counters); }
void bfrefGetCounters(BfrefCounters *out) { *out = counters; }

static void *xmalloc(size_t n) {
  ++counters.mallocs;
  void *p = malloc(n ? n : 1);
  if (!p) setError(BFABI_ERROR_MEMORY_ERROR);
  return p;
}
static void *xcalloc(size_t n, size_t sz) {
  ++counters.mallocs;
  void *p = calloc(n ? n : 1, sz);
  if (!p) setError(BFABI_ERROR_MEMORY_ERROR);
  return p;
}

/* ---- optional CBLAS backend ------------------------------------------------ */
typedef void (*zgemm_fn)(int order, int ta, int tb, int m, int n, int k, const void *alpha,
                         const void *a, int lda, const void *b, int ldb, const void *beta,
                         void *c, int ldc);
typedef void (*dgemv_fn)(int order, int trans, int m, int n, double alpha, const double *a,
                         int lda, const double *x, int incx, double beta, double *y, int incy);
static zgemm_fn blas_zgemm = NULL;
static dgemv_fn blas_dgemv = NULL;
static char blasName[512] = "builtin-c";

int bfrefUseBlas(char const *path, char const *symbolPrefix) {
  void *h = dlopen(path, RTLD_NOW | RTLD_LOCAL);
  if (!h) return 1;
  char sym[128];
  snprintf(sym, sizeof sym, "%scblas_zgemm", symbolPrefix ? symbolPrefix : "");
  zgemm_fn z = (zgemm_fn)dlsym(h, sym);
  snprintf(sym, sizeof sym, "%scblas_dgemv", symbolPrefix ? symbolPrefix : "");
  dgemv_fn d = (dgemv_fn)dlsym(h, sym);
  if (!z || !d) { dlclose(h); return 2; }
  blas_zgemm = z;
  blas_dgemv = d;
  snprintf(blasName, sizeof blasName, "%s", path);
  return 0;
}
char const *bfrefBlasName(void) { return blasName; }

/* ---- generic dispatch (src/mat.c:43-189: one-line virtual calls) ---------- */
#define SLOT(mat, idx, T) ((T)(mat)->vtbl->slot[idx])

BfMat *bfMatMul(BfMat const *lhs, BfMat const *rhs) {
  return SLOT(lhs, BFABI_SLOT_Mul, BfAbiMulFn)(lhs, rhs);
}
BfVec *bfMatMulVec(BfMat const *lhs, BfVec const *rhs) {
  return SLOT(lhs, BFABI_SLOT_MulVec, BfAbiMulVecFn)(lhs, rhs);
}
BfVec *bfMatRmulVec(BfMat const *lhs, BfVec const *rhs) {
  return SLOT(lhs, BFABI_SLOT_RmulVec, BfAbiMulVecFn)(lhs, rhs);
}
/* src/mat.c:195-197: bfMatRmul(mat, otherMat) = otherMat * mat.  (A type without the slot -- every block type: the reference's own
 * chain ends in a NULL call at BlockCoo / BlockDiag -- is an error here.) */
BfMat *bfMatRmul(BfMat const *mat, BfMat const *otherMat) {
  BfAbiMulFn f = SLOT(mat, BFABI_SLOT_Rmul, BfAbiMulFn);
  if (!f) { setError(BFABI_ERROR_NOT_IMPLEMENTED); return NULL; }
  return f(mat, otherMat);
}
void bfMatDelete(BfMat **mat) {
  if (mat && *mat) SLOT(*mat, BFABI_SLOT_Delete, BfAbiDeleteFn)(mat);
}
/* src/mat.c:271-273: in-place transposition is one more virtual call (the types restated here do not fill the slot;
 * the dispatcher is what a foreign operator behind the vtable is driven through) */
/* src/mat.c:271-273.  (A type without the slot -- BlockCoo, DenseReal -- is a NULL call in the reference; an error here.) */
void bfMatTranspose(BfMat *mat) {
  BfAbiTransposeFn f = SLOT(mat, BFABI_SLOT_Transpose, BfAbiTransposeFn);
  if (!f) { setError(BFABI_ERROR_NOT_IMPLEMENTED); return; }
  f(mat);
}
size_t bfMatGetNumRows(BfMat const *mat) { return SLOT(mat, BFABI_SLOT_GetNumRows, BfAbiGetSizeFn)(mat); }
size_t bfMatGetNumCols(BfMat const *mat) { return SLOT(mat, BFABI_SLOT_GetNumCols, BfAbiGetSizeFn)(mat); }
int bfMatGetType(BfMat const *mat) { return SLOT(mat, BFABI_SLOT_GetType, BfAbiGetTypeFn)(mat); }
size_t bfMatNumBytes(BfMat const *mat) { return SLOT(mat, BFABI_SLOT_NumBytes, BfAbiGetSizeFn)(mat); }
static BfMat *bfMatEmptyLike(BfMat const *m, size_t r, size_t c) { return SLOT(m, BFABI_SLOT_EmptyLike, BfAbiLikeFn)(m, r, c); }
static BfMat *bfMatZerosLike(BfMat const *m, size_t r, size_t c) { return SLOT(m, BFABI_SLOT_ZerosLike, BfAbiLikeFn)(m, r, c); }
typedef BfMat *(*RowRangeFn)(BfMat *, size_t, size_t);
typedef void (*SetRowRangeFn)(BfMat *, size_t, size_t, BfMat const *);
typedef void (*AddInplaceFn)(BfMat *, BfMat const *);
static BfMat *bfMatGetRowRange(BfMat *m, size_t i0, size_t i1) { return SLOT(m, BFABI_SLOT_GetRowRange, RowRangeFn)(m, i0, i1); }
static void bfMatSetRowRange(BfMat *m, size_t i0, size_t i1, BfMat const *rows) { SLOT(m, BFABI_SLOT_SetRowRange, SetRowRangeFn)(m, i0, i1, rows); }
static void bfMatAddInplace(BfMat *m, BfMat const *o) { SLOT(m, BFABI_SLOT_AddInplace, AddInplaceFn)(m, o); }

/* =========================================================================
 * Vec (real): src/vec_real.c
 * ========================================================================= */
static BfAbiVecVtable VecRealVtable;

static int vecRealGetType(BfVec const *v) { (void)v; return BFABI_TYPE_VEC_REAL; }
static void vecRealDelete(BfVec **vec) {
  BfAbiVecReal *v = (BfAbiVecReal *)*vec;
  if (!(v->super.props & 1)) free(v->data);
  free(v);
  *vec = NULL;
}
static BfVec *vecRealNewEmpty(size_t n) {        /* bfVecRealNewEmpty */
  BfAbiVecReal *v = xmalloc(sizeof *v);
  v->super.vtbl = &VecRealVtable;
  v->super.props = 0;
  v->super.size = n;
  v->stride = 1;
  v->data = xmalloc(n * sizeof(double));
  return &v->super;
}
static BfVec *vecRealNewWithValue(size_t n, double value) {
  BfVec *v = vecRealNewEmpty(n);
  double *d = ((BfAbiVecReal *)v)->data;
  for (size_t i = 0; i < n; ++i) d[i] = value;
  return v;
}
static BfVec *vecRealCopy(BfVec const *vec) {    /* src/vec_real.c bfVecRealCopy */
  BfAbiVecReal const *s = (BfAbiVecReal const *)vec;
  BfVec *c = vecRealNewEmpty(vec->size);
  double *d = ((BfAbiVecReal *)c)->data;
  for (size_t i = 0; i < vec->size; ++i) d[i] = s->data[i * s->stride];
  return c;
}
/* src/vec_real.c:112-153 bfVecRealGetSubvecView(Const): a malloc'd view */
static BfVec *vecRealGetSubvecView(BfVec *vec, size_t i0, size_t i1) {
  BfAbiVecReal *s = (BfAbiVecReal *)vec;
  BfAbiVecReal *v = xmalloc(sizeof *v);
  v->super.vtbl = &VecRealVtable;
  v->super.props = 1; /* BF_VEC_PROPS_VIEW */
  v->super.size = i1 - i0;
  v->stride = s->stride;
  v->data = s->data + i0 * s->stride;
  return &v->super;
}
/* src/vec_real.c:154 bfVecRealSetRange */
static void vecRealSetRange(BfVec *vec, size_t i0, size_t i1, BfVec const *other) {
  BfAbiVecReal *d = (BfAbiVecReal *)vec;
  BfAbiVecReal const *s = (BfAbiVecReal const *)other;
  if (i0 > i1 || i1 > vec->size) { setError(BFABI_ERROR_INVALID_ARGUMENTS); return; }
  for (size_t i = i0; i < i1; ++i) d->data[i * d->stride] = s->data[(i - i0) * s->stride];
}
/* src/vec_real.c:274 bfVecRealAddInplace */
static void vecRealAddInplace(BfVec *vec, BfVec const *other) {
  BfAbiVecReal *d = (BfAbiVecReal *)vec;
  BfAbiVecReal const *s = (BfAbiVecReal const *)other;
  if (vec->size != other->size) { setError(BFABI_ERROR_INVALID_ARGUMENTS); return; }
  for (size_t i = 0; i < vec->size; ++i) d->data[i * d->stride] += s->data[i * s->stride];
}
void bfVecDelete(BfVec **vec) {
  if (vec && *vec) ((BfAbiVecDeleteFn)(*vec)->vtbl->slot[BFABI_VSLOT_Delete])(vec);
}
BfVec *bfVecRealNewFromPtr(size_t n, double *data, int policy) {
  BfAbiVecReal *v = xmalloc(sizeof *v);
  v->super.vtbl = &VecRealVtable;
  v->super.props = policy == 1 ? 1 : 0;
  v->super.size = n;
  v->stride = 1;
  if (policy == 0) {
    v->data = xmalloc(n * sizeof(double));
    memcpy(v->data, data, n * sizeof(double));
  } else {
    v->data = data;
  }
  return &v->super;
}
double *bfVecRealData(BfVec *vec) { return ((BfAbiVecReal *)vec)->data; }

static BfAbiVecVtable VecRealVtable = {.slot = {
  [BFABI_VSLOT_Copy] = (void *)vecRealCopy,
  [BFABI_VSLOT_Delete] = (void *)vecRealDelete,
  [BFABI_VSLOT_GetType] = (void *)vecRealGetType,
  [BFABI_VSLOT_GetSubvecView] = (void *)vecRealGetSubvecView,
  [BFABI_VSLOT_GetSubvecViewConst] = (void *)vecRealGetSubvecView,
  [BFABI_VSLOT_SetRange] = (void *)vecRealSetRange,
  [BFABI_VSLOT_AddInplace] = (void *)vecRealAddInplace,
}};

/* =========================================================================
 * MatDenseComplex: src/mat_dense_complex.c
 * ========================================================================= */
static BfAbiMatVtable MatDenseComplexVtable;

static int denseComplexGetType(BfMat const *m) { (void)m; return BFABI_TYPE_MAT_DENSE_COMPLEX; }
static size_t denseGetNumRows(BfMat const *m) { return (m->props & BFABI_MAT_PROPS_TRANS) ? m->numCols : m->numRows; }
static size_t denseGetNumCols(BfMat const *m) { return (m->props & BFABI_MAT_PROPS_TRANS) ? m->numRows : m->numCols; }
/* :452-455 */
static size_t denseComplexNumBytes(BfMat const *m) { return sizeof(cplx) * m->numRows * m->numCols; }

/* bfMatDenseComplexInit :2069-2087: contiguous row-major, rowStride = numCols */
static BfAbiMatDenseComplex *denseComplexNewInit(size_t m, size_t n, int zero) {
  BfAbiMatDenseComplex *d = xmalloc(sizeof *d);
  d->super.vtbl = &MatDenseComplexVtable;
  d->super.props = 0;
  d->super.numRows = m;
  d->super.numCols = n;
  d->rowStride = n;
  d->colStride = 1;
  d->data = zero ? xcalloc(m * n, sizeof(cplx)) : xmalloc(m * n * sizeof(cplx));
  d->pyArray = NULL;
  return d;
}
/* :404-445 */
static BfMat *denseComplexEmptyLike(BfMat const *m, size_t r, size_t c) {
  if (r == (size_t)-1) r = bfMatGetNumRows(m);
  if (c == (size_t)-1) c = bfMatGetNumCols(m);
  return &denseComplexNewInit(r, c, 0)->super;
}
static BfMat *denseComplexZerosLike(BfMat const *m, size_t r, size_t c) {
  if (r == (size_t)-1) r = bfMatGetNumRows(m);
  if (c == (size_t)-1) c = bfMatGetNumCols(m);
  return &denseComplexNewInit(r, c, 1)->super;
}
/* :2164-2187 Deinit: free(data) unless VIEW, then free(struct) */
static void denseComplexDelete(BfMat **mat) {
  BfAbiMatDenseComplex *d = (BfAbiMatDenseComplex *)*mat;
  if (!(d->super.props & BFABI_MAT_PROPS_VIEW)) free(d->data);
  free(d);
  *mat = NULL;
}
/* :224-242 GetView: struct copy + VIEW flag */
static BfMat *denseComplexGetView(BfMat *mat) {
  BfAbiMatDenseComplex *v = xmalloc(sizeof *v);
  *v = *(BfAbiMatDenseComplex *)mat;
  v->super.props |= BFABI_MAT_PROPS_VIEW;
  return &v->super;
}
/* :623-646 GetRowRange: view with shifted data pointer */
static BfMat *denseComplexGetRowRange(BfMat *mat, size_t i0, size_t i1) {
  if (!(i0 < i1) || i1 > mat->numRows || (mat->props & BFABI_MAT_PROPS_TRANS)) {
    setError(BFABI_ERROR_INVALID_ARGUMENTS);
    return NULL;
  }
  BfAbiMatDenseComplex *v = (BfAbiMatDenseComplex *)denseComplexGetView(mat);
  if (i1 - i0 != mat->numRows) {
    v->super.numRows = i1 - i0;
    v->data += 2 * v->rowStride * i0;
  }
  return &v->super;
}
/* :1494-1519 Set: strided element copy */
static void denseComplexSet(BfAbiMatDenseComplex *dst, BfAbiMatDenseComplex const *src) {
  if (dst->super.numRows != src->super.numRows || dst->super.numCols != src->super.numCols) {
    setError(BFABI_ERROR_INVALID_ARGUMENTS);
    return;
  }
  cplx *D = (cplx *)dst->data;
  cplx const *S = (cplx const *)src->data;
  for (size_t i = 0; i < dst->super.numRows; ++i)
    for (size_t j = 0; j < dst->super.numCols; ++j)
      D[i * dst->rowStride + j * dst->colStride] = S[i * src->rowStride + j * src->colStride];
}
/* :718-739 SetRowRange */
static void denseComplexSetRowRange(BfMat *mat, size_t i0, size_t i1, BfMat const *rows) {
  if (bfMatGetType(rows) != BFABI_TYPE_MAT_DENSE_COMPLEX) { setError(BFABI_ERROR_NOT_IMPLEMENTED); return; }
  BfMat *view = denseComplexGetRowRange(mat, i0, i1);
  if (!view) return;
  denseComplexSet((BfAbiMatDenseComplex *)view, (BfAbiMatDenseComplex const *)rows);
  denseComplexDelete(&view);
}
/* :888-915 -> :1571-1588 AddInplace: flat loop over numRows*numCols, ignores strides */
static void denseComplexAddInplace(BfMat *mat, BfMat const *other) {
  if (bfMatGetType(other) != BFABI_TYPE_MAT_DENSE_COMPLEX) { setError(BFABI_ERROR_NOT_IMPLEMENTED); return; }
  BfAbiMatDenseComplex *a = (BfAbiMatDenseComplex *)mat;
  BfAbiMatDenseComplex const *b = (BfAbiMatDenseComplex const *)other;
  if (mat->numRows != other->numRows || mat->numCols != other->numCols) { setError(BFABI_ERROR_INVALID_ARGUMENTS); return; }
  cplx *A = (cplx *)a->data;
  cplx const *B = (cplx const *)b->data;
  size_t n = mat->numRows * mat->numCols;
  for (size_t i = 0; i < n; ++i) A[i] += B[i];
}

/* built-in row-major C[m x n] = A[m x k] * B[k x n] (what cblas_zgemm computes
 * at :1754 with alpha = 1, beta = 0, NoTrans/NoTrans) */
static void zgemm_builtin(size_t m, size_t n, size_t k, cplx const *A, size_t lda,
                          cplx const *B, size_t ldb, cplx *C, size_t ldc) {
  if (n == 1) {
    for (size_t i = 0; i < m; ++i) {
      double re = 0, im = 0;
      double const *a = (double const *)(A + i * lda);
      double const *b = (double const *)B;
      for (size_t p = 0; p < k; ++p) {
        double ar = a[2 * p], ai = a[2 * p + 1];
        double br = b[2 * p * ldb], bi = b[2 * p * ldb + 1];
        re += ar * br - ai * bi;
        im += ar * bi + ai * br;
      }
      C[i * ldc] = re + im * I;
    }
    return;
  }
  for (size_t i = 0; i < m; ++i) {
    double *c = (double *)(C + i * ldc);
    for (size_t j = 0; j < 2 * n; ++j) c[j] = 0;
    for (size_t p = 0; p < k; ++p) {
      double ar = creal(A[i * lda + p]), ai = cimag(A[i * lda + p]);
      double const *b = (double const *)(B + p * ldb);
      for (size_t j = 0; j < n; ++j) {
        c[2 * j] += ar * b[2 * j] - ai * b[2 * j + 1];
        c[2 * j + 1] += ar * b[2 * j + 1] + ai * b[2 * j];
      }
    }
  }
}

/* built-in row-major C[m x n] = A^H B with A stored k x m (what cblas_zgemm computes at :1754 for CblasConjTrans) */
static void zgemm_builtin_conjtrans(size_t m, size_t n, size_t k, cplx const *A, size_t lda,
                                    cplx const *B, size_t ldb, cplx *C, size_t ldc) {
  for (size_t i = 0; i < m; ++i) {
    double *c = (double *)(C + i * ldc);
    for (size_t j = 0; j < 2 * n; ++j) c[j] = 0;
    for (size_t p = 0; p < k; ++p) {
      double ar = creal(A[p * lda + i]), ai = -cimag(A[p * lda + i]);
      double const *b = (double const *)(B + p * ldb);
      for (size_t j = 0; j < n; ++j) {
        c[2 * j] += ar * b[2 * j] - ai * b[2 * j + 1];
        c[2 * j + 1] += ar * b[2 * j + 1] + ai * b[2 * j];
      }
    }
  }
}

/* :1024-1051 -> :1704-1765: result = new m x n; zgemm(alpha=1, beta=0).
 * getCblasTranspose (:27-35) maps TRANS *or* CONJ to CblasConjTrans (its `else if` for a plain CblasTrans can never be
 * reached): a left operand that bfMatDenseComplexTranspose (:1475-1478, = bfMatConjTrans, src/mat.c:359-362) has flagged
 * multiplies as its conjugate transpose, with the extents bfMatGetNumRows / GetNumCols report (:503-511, swapped under
 * TRANS).  CONJ without TRANS would hand zgemm ConjTrans with UNswapped extents -- nothing in the reference produces
 * that state; refused here.  A flagged right operand is not on the apply path; refused. */
static BfMat *denseComplexMul(BfMat const *op1, BfMat const *op2) {
  if (bfMatGetNumCols(op1) != bfMatGetNumRows(op2)) { setError(BFABI_ERROR_INVALID_ARGUMENTS); return NULL; }
  if (bfMatGetType(op2) != BFABI_TYPE_MAT_DENSE_COMPLEX) { setError(BFABI_ERROR_NOT_IMPLEMENTED); return NULL; }
  if (op2->props & (BFABI_MAT_PROPS_TRANS | BFABI_MAT_PROPS_CONJ)) { setError(BFABI_ERROR_NOT_IMPLEMENTED); return NULL; }
  int const flagged = (op1->props & (BFABI_MAT_PROPS_TRANS | BFABI_MAT_PROPS_CONJ)) != 0;
  if (flagged && !(op1->props & BFABI_MAT_PROPS_TRANS)) { setError(BFABI_ERROR_NOT_IMPLEMENTED); return NULL; }
  BfAbiMatDenseComplex const *a = (BfAbiMatDenseComplex const *)op1;
  BfAbiMatDenseComplex const *b = (BfAbiMatDenseComplex const *)op2;
  size_t m = bfMatGetNumRows(op1), k = bfMatGetNumCols(op1), n = op2->numCols;
  if (!(m > 0 && n > 0 && k > 0)) { setError(BFABI_ERROR_INVALID_ARGUMENTS); return NULL; }
  BfAbiMatDenseComplex *c = denseComplexNewInit(m, n, 0);
  ++counters.gemmCalls;
  counters.macs += (uint64_t)m * n * k;
  if (blas_zgemm) {
    double alpha[2] = {1, 0}, beta[2] = {0, 0};
    blas_zgemm(101 /*RowMajor*/, flagged ? 113 /*ConjTrans*/ : 111, 111, (int)m, (int)n, (int)k, alpha, a->data, (int)a->rowStride,
               b->data, (int)b->rowStride, beta, c->data, (int)c->rowStride);
  } else if (flagged) {
    zgemm_builtin_conjtrans(m, n, k, (cplx const *)a->data, a->rowStride, (cplx const *)b->data, b->rowStride,
                            (cplx *)c->data, c->rowStride);
  } else {
    zgemm_builtin(m, n, k, (cplx const *)a->data, a->rowStride, (cplx const *)b->data, b->rowStride,
                  (cplx *)c->data, c->rowStride);
  }
  return &c->super;
}
/* :1075-1133 Rmul: result (m x n) = other (m x k) * this (k x n), one zgemm; a dense complex `other` only (:1125-1133).  Flagged
 * operands are not on this path; refused. */
static BfMat *denseComplexRmul(BfMat const *mat, BfMat const *other) {
  if (bfMatGetType(other) != BFABI_TYPE_MAT_DENSE_COMPLEX) { setError(BFABI_ERROR_NOT_IMPLEMENTED); return NULL; }
  if ((mat->props | other->props) & (BFABI_MAT_PROPS_TRANS | BFABI_MAT_PROPS_CONJ)) { setError(BFABI_ERROR_NOT_IMPLEMENTED); return NULL; }
  size_t const m = bfMatGetNumRows(other), n = bfMatGetNumCols(mat), k = bfMatGetNumCols(other);
  if (k != bfMatGetNumRows(mat)) { setError(BFABI_ERROR_INCOMPATIBLE_SHAPES); return NULL; }
  if (!(m > 0 && n > 0 && k > 0)) { setError(BFABI_ERROR_INVALID_ARGUMENTS); return NULL; }
  BfAbiMatDenseComplex const *a = (BfAbiMatDenseComplex const *)other, *b = (BfAbiMatDenseComplex const *)mat;
  BfAbiMatDenseComplex *c = denseComplexNewInit(m, n, 0);
  ++counters.gemmCalls;
  counters.macs += (uint64_t)m * n * k;
  if (blas_zgemm) {
    double alpha[2] = {1, 0}, beta[2] = {0, 0};
    blas_zgemm(101 /*RowMajor*/, 111, 111, (int)m, (int)n, (int)k, alpha, a->data, (int)a->rowStride, b->data, (int)b->rowStride, beta, c->data, (int)c->rowStride);
  } else
    zgemm_builtin(m, n, k, (cplx const *)a->data, a->rowStride, (cplx const *)b->data, b->rowStride, (cplx *)c->data, c->rowStride);
  return &c->super;
}
/* :1475-1478 */
static void denseComplexTranspose(BfMat *mat) { mat->props ^= (BFABI_MAT_PROPS_TRANS | BFABI_MAT_PROPS_CONJ); }

static BfAbiMatVtable MatDenseComplexVtable = {.slot = {
  [BFABI_SLOT_GetView] = (void *)denseComplexGetView,
  [BFABI_SLOT_Delete] = (void *)denseComplexDelete,
  [BFABI_SLOT_EmptyLike] = (void *)denseComplexEmptyLike,
  [BFABI_SLOT_ZerosLike] = (void *)denseComplexZerosLike,
  [BFABI_SLOT_GetType] = (void *)denseComplexGetType,
  [BFABI_SLOT_NumBytes] = (void *)denseComplexNumBytes,
  [BFABI_SLOT_GetNumRows] = (void *)denseGetNumRows,
  [BFABI_SLOT_GetNumCols] = (void *)denseGetNumCols,
  [BFABI_SLOT_GetRowRange] = (void *)denseComplexGetRowRange,
  [BFABI_SLOT_SetRowRange] = (void *)denseComplexSetRowRange,
  [BFABI_SLOT_AddInplace] = (void *)denseComplexAddInplace,
  [BFABI_SLOT_Mul] = (void *)denseComplexMul,
  [BFABI_SLOT_Rmul] = (void *)denseComplexRmul,
  [BFABI_SLOT_Transpose] = (void *)denseComplexTranspose,
}};

BfMat *bfMatDenseComplexNewFromPtr(size_t m, size_t n, double *data, int policy) {
  BfAbiMatDenseComplex *d = xmalloc(sizeof *d);
  d->super.vtbl = &MatDenseComplexVtable;
  d->super.props = policy == 1 ? BFABI_MAT_PROPS_VIEW : 0;
  d->super.numRows = m;
  d->super.numCols = n;
  d->rowStride = n;
  d->colStride = 1;
  d->pyArray = NULL;
  if (policy == 0) {
    d->data = xmalloc(m * n * sizeof(cplx));
    memcpy(d->data, data, m * n * sizeof(cplx));
  } else {
    d->data = data;
  }
  return &d->super;
}
BfMat *bfMatDenseComplexNewZeros(size_t m, size_t n) { return &denseComplexNewInit(m, n, 1)->super; }

/* =========================================================================
 * MatDenseReal: src/mat_dense_real.c
 * ========================================================================= */
static BfAbiMatVtable MatDenseRealVtable;
static int denseRealGetType(BfMat const *m) { (void)m; return BFABI_TYPE_MAT_DENSE_REAL; }
static size_t denseRealNumBytes(BfMat const *m) { return sizeof(double) * m->numRows * m->numCols; }
static void denseRealDelete(BfMat **mat) {
  BfAbiMatDenseReal *d = (BfAbiMatDenseReal *)*mat;
  if (!(d->super.super.props & BFABI_MAT_PROPS_VIEW)) free(d->data);
  free(d);
  *mat = NULL;
}
/* :1373-1407 -> mulVec_vecReal :1340-1371: result = new vec(m); dgemv(alpha=1,beta=0) */
static BfVec *denseRealMulVec(BfMat const *mat, BfVec const *vec) {
  BfAbiMatDenseReal const *a = (BfAbiMatDenseReal const *)mat;
  size_t m = bfMatGetNumRows(mat), n = bfMatGetNumCols(mat);
  if (n != vec->size || n == 0) { setError(BFABI_ERROR_INVALID_ARGUMENTS); return NULL; }
  if (((BfAbiVecGetTypeFn)vec->vtbl->slot[BFABI_VSLOT_GetType])(vec) != BFABI_TYPE_VEC_REAL) { setError(BFABI_ERROR_NOT_IMPLEMENTED); return NULL; }
  if (mat->props & (BFABI_MAT_PROPS_TRANS | BFABI_MAT_PROPS_CONJ)) { setError(BFABI_ERROR_NOT_IMPLEMENTED); return NULL; }
  BfAbiVecReal const *x = (BfAbiVecReal const *)vec;
  BfVec *res = vecRealNewEmpty(m);
  double *y = ((BfAbiVecReal *)res)->data;
  ++counters.gemmCalls;
  counters.macs += (uint64_t)m * n;
  if (blas_dgemv) {
    blas_dgemv(101, 111, (int)m, (int)n, 1.0, a->data, (int)a->super.rowStride, x->data, (int)x->stride, 0.0, y, 1);
  } else {
    for (size_t i = 0; i < m; ++i) {
      double const *row = a->data + i * a->super.rowStride;
      double s = 0;
      for (size_t j = 0; j < n; ++j) s += row[j * a->super.colStride] * x->data[j * x->stride];
      y[i] = s;
    }
  }
  return res;
}
/* :1508-1542 -> rmulVec_vecReal :1410-1440: result = new vec(n); dgemv with the opposite transpose */
static BfVec *denseRealRmulVec(BfMat const *mat, BfVec const *vec) {
  BfAbiMatDenseReal const *a = (BfAbiMatDenseReal const *)mat;
  size_t m = bfMatGetNumRows(mat), n = bfMatGetNumCols(mat);
  if (m != vec->size || m == 0) { setError(BFABI_ERROR_INVALID_ARGUMENTS); return NULL; }
  if (((BfAbiVecGetTypeFn)vec->vtbl->slot[BFABI_VSLOT_GetType])(vec) != BFABI_TYPE_VEC_REAL) { setError(BFABI_ERROR_NOT_IMPLEMENTED); return NULL; }
  if (mat->props & (BFABI_MAT_PROPS_TRANS | BFABI_MAT_PROPS_CONJ)) { setError(BFABI_ERROR_NOT_IMPLEMENTED); return NULL; }
  BfAbiVecReal const *x = (BfAbiVecReal const *)vec;
  BfVec *res = vecRealNewEmpty(n);
  double *y = ((BfAbiVecReal *)res)->data;
  ++counters.gemmCalls;
  counters.macs += (uint64_t)m * n;
  if (blas_dgemv) {
    blas_dgemv(101, 112 /*CblasTrans*/, (int)m, (int)n, 1.0, a->data, (int)a->super.rowStride, x->data, (int)x->stride, 0.0, y, 1);
  } else {
    for (size_t j = 0; j < n; ++j) y[j] = 0;
    for (size_t i = 0; i < m; ++i) {
      double const *row = a->data + i * a->super.rowStride;
      double xi = x->data[i * x->stride];
      for (size_t j = 0; j < n; ++j) y[j] += row[j * a->super.colStride] * xi;
    }
  }
  return res;
}
static BfAbiMatVtable MatDenseRealVtable = {.slot = {
  [BFABI_SLOT_RmulVec] = (void *)denseRealRmulVec,
  [BFABI_SLOT_Delete] = (void *)denseRealDelete,
  [BFABI_SLOT_GetType] = (void *)denseRealGetType,
  [BFABI_SLOT_NumBytes] = (void *)denseRealNumBytes,
  [BFABI_SLOT_GetNumRows] = (void *)denseGetNumRows,
  [BFABI_SLOT_GetNumCols] = (void *)denseGetNumCols,
  [BFABI_SLOT_MulVec] = (void *)denseRealMulVec,
}};
BfMat *bfMatDenseRealNewFromPtr(size_t m, size_t n, double *data, int policy) {
  BfAbiMatDenseReal *d = xmalloc(sizeof *d);
  d->super.super.vtbl = &MatDenseRealVtable;
  d->super.super.props = policy == 1 ? BFABI_MAT_PROPS_VIEW : 0;
  d->super.super.numRows = m;
  d->super.super.numCols = n;
  d->super.vtable = NULL;
  d->super.rowStride = n;
  d->super.colStride = 1;
  if (policy == 0) {
    d->data = xmalloc(m * n * sizeof(double));
    memcpy(d->data, data, m * n * sizeof(double));
  } else {
    d->data = data;
  }
  return &d->super.super;
}
double *bfMatDenseData(BfMat *mat) {
  int t = bfMatGetType(mat);
  if (t == BFABI_TYPE_MAT_DENSE_COMPLEX) return ((BfAbiMatDenseComplex *)mat)->data;
  if (t == BFABI_TYPE_MAT_DENSE_REAL) return ((BfAbiMatDenseReal *)mat)->data;
  return NULL;
}

/* =========================================================================
 * MatIdentity: src/mat_identity.c:149-181
 * ========================================================================= */
static BfAbiMatVtable MatIdentityVtable;
static int identityGetType(BfMat const *m) { (void)m; return BFABI_TYPE_MAT_IDENTITY; }
static size_t plainGetNumRows(BfMat const *m) { return m->numRows; }
static size_t plainGetNumCols(BfMat const *m) { return m->numCols; }
static size_t identityNumBytes(BfMat const *m) { (void)m; return 0; }
static void identityDelete(BfMat **mat) { free(*mat); *mat = NULL; }
/* :149-163 Mul: square only, result = copy(rhs) */
static BfMat *identityMul(BfMat const *mat, BfMat const *rhs) {
  if (mat->numRows != mat->numCols) { setError(BFABI_ERROR_NOT_IMPLEMENTED); return NULL; }
  if (bfMatGetType(rhs) != BFABI_TYPE_MAT_DENSE_COMPLEX) { setError(BFABI_ERROR_NOT_IMPLEMENTED); return NULL; }
  BfMat *c = bfMatEmptyLike(rhs, rhs->numRows, rhs->numCols);
  denseComplexSet((BfAbiMatDenseComplex *)c, (BfAbiMatDenseComplex const *)rhs);
  return c;
}
/* :165-181 MulVec: square only, result = copy(vec) */
static BfVec *identityMulVec(BfMat const *mat, BfVec const *vec) {
  if (mat->numRows != mat->numCols) { setError(BFABI_ERROR_NOT_IMPLEMENTED); return NULL; }
  return ((BfAbiVecCopyFn)vec->vtbl->slot[BFABI_VSLOT_Copy])(vec);
}
/* src/mat_identity.c:183-198 RmulVec: square only, result = copy(vec) */
static BfVec *identityRmulVec(BfMat const *mat, BfVec const *vec) { return identityMulVec(mat, vec); }
/* src/mat_identity.c:210-212 */
static void identityTranspose(BfMat *mat) { size_t const t = mat->numRows; mat->numRows = mat->numCols; mat->numCols = t; }
static BfAbiMatVtable MatIdentityVtable = {.slot = {
  [BFABI_SLOT_Transpose] = (void *)identityTranspose,
  [BFABI_SLOT_RmulVec] = (void *)identityRmulVec,
  [BFABI_SLOT_Delete] = (void *)identityDelete,
  [BFABI_SLOT_GetType] = (void *)identityGetType,
  [BFABI_SLOT_NumBytes] = (void *)identityNumBytes,
  [BFABI_SLOT_GetNumRows] = (void *)plainGetNumRows,
  [BFABI_SLOT_GetNumCols] = (void *)plainGetNumCols,
  [BFABI_SLOT_Mul] = (void *)identityMul,
  [BFABI_SLOT_MulVec] = (void *)identityMulVec,
}};
BfMat *bfMatIdentityNew(size_t n) {
  BfAbiMatIdentity *m = xmalloc(sizeof *m);
  m->super.vtbl = &MatIdentityVtable;
  m->super.props = 0;
  m->super.numRows = m->super.numCols = n;
  return &m->super;
}

/* =========================================================================
 * MatBlock family: src/mat_block.c, mat_block_{diag,coo,dense}.c
 * super.numRows/numCols = block counts (mat_block.c:104)
 * ========================================================================= */
static size_t blockGetNumRows(BfMat const *m) { BfAbiMatBlock const *b = (BfAbiMatBlock const *)m; return b->rowOffset[m->numRows]; }
static size_t blockGetNumCols(BfMat const *m) { BfAbiMatBlock const *b = (BfAbiMatBlock const *)m; return b->colOffset[m->numCols]; }

static void blockFreeCommon(BfAbiMatBlock *b, size_t numBlocks) {
  if (!(b->super.props & BFABI_MAT_PROPS_VIEW)) {
    for (size_t k = 0; k < numBlocks; ++k) bfMatDelete(&b->block[k]);
    free(b->block);
    free(b->rowOffset);
    free(b->colOffset);
  }
}
static void blockInitCommon(BfAbiMatBlock *b, BfAbiMatVtable *vt, size_t nbr, size_t nbc, size_t numBlocks,
                            size_t const *rowOffset, size_t const *colOffset, BfMat **blocks) {
  b->super.vtbl = vt;
  b->super.props = 0;
  b->super.numRows = nbr;
  b->super.numCols = nbc;
  b->vtbl = NULL;
  b->block = xmalloc(numBlocks * sizeof(BfMat *));
  memcpy(b->block, blocks, numBlocks * sizeof(BfMat *));
  b->rowOffset = xmalloc((nbr + 1) * sizeof(size_t));
  memcpy(b->rowOffset, rowOffset, (nbr + 1) * sizeof(size_t));
  b->colOffset = xmalloc((nbc + 1) * sizeof(size_t));
  memcpy(b->colOffset, colOffset, (nbc + 1) * sizeof(size_t));
}

/* ---- BlockDiag ------------------------------------------------------------ */
static BfAbiMatVtable MatBlockDiagVtable;
static int blockDiagGetType(BfMat const *m) { (void)m; return BFABI_TYPE_MAT_BLOCK_DIAG; }
static size_t blockDiagNumBlocks(BfMat const *m) { return m->numRows < m->numCols ? m->numRows : m->numCols; }
static size_t blockDiagNumBytes(BfMat const *m) {
  BfAbiMatBlock const *b = (BfAbiMatBlock const *)m;
  size_t n = 0;
  for (size_t k = 0; k < blockDiagNumBlocks(m); ++k) n += bfMatNumBytes(b->block[k]);
  return n;
}
static void blockDiagDelete(BfMat **mat) {
  BfAbiMatBlock *b = (BfAbiMatBlock *)*mat;
  blockFreeCommon(b, blockDiagNumBlocks(*mat));
  free(b);
  *mat = NULL;
}
/* src/mat_block_diag.c:370-405 */
static BfMat *blockDiagMul(BfMat const *mat, BfMat const *other) {
  BfAbiMatBlock const *matBlock = (BfAbiMatBlock const *)mat;
  size_t numRows = bfMatGetNumRows(mat);
  size_t numCols = bfMatGetNumCols(other);
  size_t numBlocks = blockDiagNumBlocks(mat);

  BfMat *result = bfMatEmptyLike(other, numRows, numCols);
  if (currentError) { bfMatDelete(&result); return NULL; }

  for (size_t i = 0, i0, i1, j0, j1; i < numBlocks; ++i) {
    i0 = matBlock->rowOffset[i];
    i1 = matBlock->rowOffset[i + 1];
    j0 = matBlock->colOffset[i];
    j1 = matBlock->colOffset[i + 1];

    BfMat *op2Rows = bfMatGetRowRange((BfMat *)other, j0, j1);
    BfMat *resultRows = op2Rows ? bfMatMul(matBlock->block[i], op2Rows) : NULL;
    if (resultRows) bfMatSetRowRange(result, i0, i1, resultRows);

    bfMatDelete(&resultRows);
    bfMatDelete(&op2Rows);
    if (currentError) { bfMatDelete(&result); return NULL; }
  }
  return result;
}
/* src/mat_block_diag.c:407-456 */
static BfVec *blockDiagMulVec(BfMat const *mat, BfVec const *vec) {
  BfAbiMatBlock const *matBlock = (BfAbiMatBlock const *)mat;
  if (bfMatGetNumCols(mat) != vec->size) { setError(BFABI_ERROR_NOT_IMPLEMENTED); return NULL; }
  if (((BfAbiVecGetTypeFn)vec->vtbl->slot[BFABI_VSLOT_GetType])(vec) != BFABI_TYPE_VEC_REAL) { setError(BFABI_ERROR_NOT_IMPLEMENTED); return NULL; }
  BfVec *result = vecRealNewEmpty(bfMatGetNumRows(mat));
  size_t numBlocks = blockDiagNumBlocks(mat);
  for (size_t k = 0; k < numBlocks; ++k) {
    BfMat const *block = matBlock->block[k];
    size_t i0 = matBlock->rowOffset[k];
    size_t i1 = i0 + bfMatGetNumRows(block);
    size_t j0 = matBlock->colOffset[k];
    size_t j1 = j0 + bfMatGetNumCols(block);
    BfVec *subvecView = vecRealGetSubvecView((BfVec *)vec, j0, j1);
    BfVec *tmp = bfMatMulVec(block, subvecView);
    bfVecDelete(&subvecView);
    if (tmp) vecRealSetRange(result, i0, i1, tmp);
    bfVecDelete(&tmp);
    if (currentError) { bfVecDelete(&result); return NULL; }
  }
  return result;
}
/* src/mat_block_diag.c:458-505 */
static BfVec *blockDiagRmulVec(BfMat const *mat, BfVec const *vec) {
  BfAbiMatBlock const *matBlock = (BfAbiMatBlock const *)mat;
  if (bfMatGetNumRows(mat) != vec->size) { setError(BFABI_ERROR_NOT_IMPLEMENTED); return NULL; }
  if (((BfAbiVecGetTypeFn)vec->vtbl->slot[BFABI_VSLOT_GetType])(vec) != BFABI_TYPE_VEC_REAL) { setError(BFABI_ERROR_NOT_IMPLEMENTED); return NULL; }
  BfVec *result = vecRealNewEmpty(bfMatGetNumCols(mat));
  size_t numBlocks = blockDiagNumBlocks(mat);
  for (size_t k = 0; k < numBlocks; ++k) {
    BfMat const *block = matBlock->block[k];
    size_t i0 = matBlock->rowOffset[k];
    size_t i1 = i0 + bfMatGetNumRows(block);
    size_t j0 = matBlock->colOffset[k];
    size_t j1 = j0 + bfMatGetNumCols(block);
    BfVec *subvecView = vecRealGetSubvecView((BfVec *)vec, i0, i1);
    BfVec *tmp = bfMatRmulVec(block, subvecView);
    bfVecDelete(&subvecView);
    if (tmp) vecRealSetRange(result, j0, j1, tmp);
    bfVecDelete(&tmp);   /* the reference leaks tmp here (:497-499) */
    if (currentError) { bfVecDelete(&result); return NULL; }
  }
  return result;
}
/* src/mat_block_diag.c:603-624: extents and offsets swapped, every diagonal block transposed in place */
static void blockDiagTranspose(BfMat *mat) {
  BfAbiMatBlock *b = (BfAbiMatBlock *)mat;
  if (mat->props & BFABI_MAT_PROPS_VIEW) { setError(BFABI_ERROR_INVALID_ARGUMENTS); return; }
  size_t const t = mat->numRows; mat->numRows = mat->numCols; mat->numCols = t;
  size_t *o = b->rowOffset; b->rowOffset = b->colOffset; b->colOffset = o;
  for (size_t k = 0; k < blockDiagNumBlocks(mat); ++k) {
    bfMatTranspose(b->block[k]);
    if (currentError) return;
  }
}
static BfAbiMatVtable MatBlockDiagVtable = {.slot = {
  [BFABI_SLOT_Transpose] = (void *)blockDiagTranspose,
  [BFABI_SLOT_RmulVec] = (void *)blockDiagRmulVec,
  [BFABI_SLOT_Delete] = (void *)blockDiagDelete,
  [BFABI_SLOT_GetType] = (void *)blockDiagGetType,
  [BFABI_SLOT_NumBytes] = (void *)blockDiagNumBytes,
  [BFABI_SLOT_GetNumRows] = (void *)blockGetNumRows,
  [BFABI_SLOT_GetNumCols] = (void *)blockGetNumCols,
  [BFABI_SLOT_Mul] = (void *)blockDiagMul,
  [BFABI_SLOT_MulVec] = (void *)blockDiagMulVec,
}};
BfMat *bfMatBlockDiagNewFromBlocks(size_t numBlocks, BfMat **blocks) {
  size_t *ro = xmalloc((numBlocks + 1) * sizeof(size_t));
  size_t *co = xmalloc((numBlocks + 1) * sizeof(size_t));
  ro[0] = co[0] = 0;
  for (size_t k = 0; k < numBlocks; ++k) {
    ro[k + 1] = ro[k] + bfMatGetNumRows(blocks[k]);
    co[k + 1] = co[k] + bfMatGetNumCols(blocks[k]);
  }
  BfAbiMatBlock *b = xmalloc(sizeof *b);
  blockInitCommon(b, &MatBlockDiagVtable, numBlocks, numBlocks, numBlocks, ro, co, blocks);
  free(ro);
  free(co);
  return &b->super;
}

/* ---- BlockCoo ------------------------------------------------------------- */
static BfAbiMatVtable MatBlockCooVtable;
static int blockCooGetType(BfMat const *m) { (void)m; return BFABI_TYPE_MAT_BLOCK_COO; }
static size_t blockCooNumBytes(BfMat const *m) {   /* src/mat_block_coo.c:238-258 */
  BfAbiMatBlockCoo const *c = (BfAbiMatBlockCoo const *)m;
  size_t n = 0;
  for (size_t k = 0; k < c->numBlocks; ++k) n += bfMatNumBytes(c->super.block[k]);
  return n;
}
static void blockCooDelete(BfMat **mat) {
  BfAbiMatBlockCoo *c = (BfAbiMatBlockCoo *)*mat;
  int view = c->super.super.props & BFABI_MAT_PROPS_VIEW;
  blockFreeCommon(&c->super, c->numBlocks);
  if (!view) { free(c->rowInd); free(c->colInd); }
  free(c);
  *mat = NULL;
}
/* src/mat_block_coo.c:382-425 */
static BfMat *blockCooMul(BfMat const *op1, BfMat const *op2) {
  BfAbiMatBlock const *matBlock1 = (BfAbiMatBlock const *)op1;
  BfAbiMatBlockCoo const *matBlockCoo1 = (BfAbiMatBlockCoo const *)op1;

  size_t numRows = bfMatGetNumRows(op1);
  size_t numCols = bfMatGetNumCols(op2);
  if (bfMatGetNumCols(op1) != bfMatGetNumRows(op2)) { setError(BFABI_ERROR_INVALID_ARGUMENTS); return NULL; }
  size_t numBlocks = matBlockCoo1->numBlocks;

  BfMat *result = bfMatZerosLike(op2, numRows, numCols);
  if (currentError) { bfMatDelete(&result); return NULL; }

  for (size_t k = 0, i0, i1, j0, j1; k < numBlocks; ++k) {
    i0 = matBlock1->rowOffset[matBlockCoo1->rowInd[k]];
    i1 = matBlock1->rowOffset[matBlockCoo1->rowInd[k] + 1];
    j0 = matBlock1->colOffset[matBlockCoo1->colInd[k]];
    j1 = matBlock1->colOffset[matBlockCoo1->colInd[k] + 1];

    BfMat *op2Rows = bfMatGetRowRange((BfMat *)op2, j0, j1);
    BfMat *tmp = op2Rows ? bfMatMul(matBlock1->block[k], op2Rows) : NULL;
    BfMat *resultRows = bfMatGetRowRange(result, i0, i1);
    if (tmp && resultRows) bfMatAddInplace(resultRows, tmp);

    bfMatDelete(&resultRows);
    bfMatDelete(&tmp);
    bfMatDelete(&op2Rows);
    if (currentError) { bfMatDelete(&result); return NULL; }
  }
  return result;
}
/* src/mat_block_coo.c:427-474 */
static BfVec *blockCooMulVec(BfMat const *mat, BfVec const *vec) {
  BfAbiMatBlock const *matBlock = (BfAbiMatBlock const *)mat;
  BfAbiMatBlockCoo const *coo = (BfAbiMatBlockCoo const *)mat;
  if (bfMatGetNumCols(mat) != vec->size) { setError(BFABI_ERROR_NOT_IMPLEMENTED); return NULL; }
  if (((BfAbiVecGetTypeFn)vec->vtbl->slot[BFABI_VSLOT_GetType])(vec) != BFABI_TYPE_VEC_REAL) { setError(BFABI_ERROR_NOT_IMPLEMENTED); return NULL; }
  BfVec *result = vecRealNewWithValue(bfMatGetNumRows(mat), 0);
  for (size_t k = 0; k < coo->numBlocks; ++k) {
    BfMat const *block = matBlock->block[k];
    size_t i0 = matBlock->rowOffset[coo->rowInd[k]];
    size_t i1 = i0 + bfMatGetNumRows(block);
    size_t j0 = matBlock->colOffset[coo->colInd[k]];
    size_t j1 = j0 + bfMatGetNumCols(block);
    BfVec *subvecView = vecRealGetSubvecView((BfVec *)vec, j0, j1);
    BfVec *tmp = bfMatMulVec(block, subvecView);
    bfVecDelete(&subvecView);
    BfVec *resultSubvecView = vecRealGetSubvecView(result, i0, i1);
    if (tmp) vecRealAddInplace(resultSubvecView, tmp);
    bfVecDelete(&resultSubvecView);
    bfVecDelete(&tmp);
    if (currentError) { bfVecDelete(&result); return NULL; }
  }
  return result;
}
/* src/mat_block_coo.c:476-520 */
static BfVec *blockCooRmulVec(BfMat const *mat, BfVec const *vec) {
  BfAbiMatBlock const *matBlock = (BfAbiMatBlock const *)mat;
  BfAbiMatBlockCoo const *coo = (BfAbiMatBlockCoo const *)mat;
  if (bfMatGetNumRows(mat) != vec->size) { setError(BFABI_ERROR_NOT_IMPLEMENTED); return NULL; }
  if (((BfAbiVecGetTypeFn)vec->vtbl->slot[BFABI_VSLOT_GetType])(vec) != BFABI_TYPE_VEC_REAL) { setError(BFABI_ERROR_NOT_IMPLEMENTED); return NULL; }
  BfVec *result = vecRealNewWithValue(bfMatGetNumCols(mat), 0);
  for (size_t k = 0; k < coo->numBlocks; ++k) {
    BfMat const *block = matBlock->block[k];
    size_t i0 = matBlock->rowOffset[coo->rowInd[k]];
    size_t i1 = i0 + bfMatGetNumRows(block);
    size_t j0 = matBlock->colOffset[coo->colInd[k]];
    size_t j1 = j0 + bfMatGetNumCols(block);
    BfVec *subvecView = vecRealGetSubvecView((BfVec *)vec, i0, i1);
    BfVec *tmp = bfMatRmulVec(block, subvecView);
    bfVecDelete(&subvecView);
    BfVec *resultSubvecView = vecRealGetSubvecView(result, j0, j1);
    if (tmp) vecRealAddInplace(resultSubvecView, tmp);
    bfVecDelete(&resultSubvecView);
    bfVecDelete(&tmp);
    if (currentError) { bfVecDelete(&result); return NULL; }
  }
  return result;
}
static BfAbiMatVtable MatBlockCooVtable = {.slot = {
  [BFABI_SLOT_RmulVec] = (void *)blockCooRmulVec,
  [BFABI_SLOT_Delete] = (void *)blockCooDelete,
  [BFABI_SLOT_GetType] = (void *)blockCooGetType,
  [BFABI_SLOT_NumBytes] = (void *)blockCooNumBytes,
  [BFABI_SLOT_GetNumRows] = (void *)blockGetNumRows,
  [BFABI_SLOT_GetNumCols] = (void *)blockGetNumCols,
  [BFABI_SLOT_Mul] = (void *)blockCooMul,
  [BFABI_SLOT_MulVec] = (void *)blockCooMulVec,
}};
BfMat *bfMatBlockCooNewFromArrays(size_t nbr, size_t nbc, size_t numBlocks, size_t const *rowOffset,
                                  size_t const *colOffset, size_t const *rowInd, size_t const *colInd,
                                  BfMat **blocks) {
  BfAbiMatBlockCoo *c = xmalloc(sizeof *c);
  blockInitCommon(&c->super, &MatBlockCooVtable, nbr, nbc, numBlocks, rowOffset, colOffset, blocks);
  c->numBlocks = numBlocks;
  c->rowInd = xmalloc(numBlocks * sizeof(size_t));
  memcpy(c->rowInd, rowInd, numBlocks * sizeof(size_t));
  c->colInd = xmalloc(numBlocks * sizeof(size_t));
  memcpy(c->colInd, colInd, numBlocks * sizeof(size_t));
  return &c->super.super;
}

/* ---- BlockDense ----------------------------------------------------------- */
static BfAbiMatVtable MatBlockDenseVtable;
static int blockDenseGetType(BfMat const *m) { (void)m; return BFABI_TYPE_MAT_BLOCK_DENSE; }
static size_t blockDenseNumBytes(BfMat const *m) {
  BfAbiMatBlock const *b = (BfAbiMatBlock const *)m;
  size_t n = 0;
  for (size_t k = 0; k < m->numRows * m->numCols; ++k) n += bfMatNumBytes(b->block[k]);
  return n;
}
static void blockDenseDelete(BfMat **mat) {
  BfAbiMatBlock *b = (BfAbiMatBlock *)*mat;
  blockFreeCommon(b, (*mat)->numRows * (*mat)->numCols);
  free(b);
  *mat = NULL;
}
/* src/mat_block_dense.c:512-572.  bfMatBlockDenseGetBlockConst (:1043-1061)
 * hands out a *view* of the block: bfMatGet(block, BF_POLICY_VIEW) ->
 * block->vtbl->GetView(block), deleted after the product (:559).  A block
 * whose vtable fills GetView (dense leaves here; a foreign operator such as
 * the device shim) goes through exactly that; this file's own container
 * types expose no GetView, so their pointer is used directly -- the
 * arithmetic is identical, one malloc/free pair per block is re-enacted by
 * the stand-in below (counted so the cpu_baseline report can state it). */
typedef BfMat *(*GetViewFn)(BfMat *);
static BfMat *blockDenseMul(BfMat const *mat, BfMat const *otherMat) {
  BfAbiMatBlock const *matBlock = (BfAbiMatBlock const *)mat;
  if (bfMatGetNumCols(mat) != bfMatGetNumRows(otherMat)) { setError(BFABI_ERROR_INCOMPATIBLE_SHAPES); return NULL; }
  size_t numRowBlocks = mat->numRows, numColBlocks = mat->numCols;
  size_t numRows = bfMatGetNumRows(mat), numCols = bfMatGetNumCols(otherMat);

  BfMat *result = bfMatZerosLike(otherMat, numRows, numCols);
  if (currentError) { bfMatDelete(&result); return NULL; }

  for (size_t i = 0, i0, i1; i < numRowBlocks; ++i) {
    i0 = matBlock->rowOffset[i];
    i1 = matBlock->rowOffset[i + 1];
    BfMat *resultRows = bfMatGetRowRange(result, i0, i1);
    if (currentError) { bfMatDelete(&resultRows); bfMatDelete(&result); return NULL; }

    for (size_t j = 0, j0, j1; j < numColBlocks; ++j) {
      j0 = matBlock->colOffset[j];
      j1 = matBlock->colOffset[j + 1];
      BfMat *op2Rows = bfMatGetRowRange((BfMat *)otherMat, j0, j1);
      BfMat const *block = matBlock->block[i * numColBlocks + j];
      GetViewFn getView = (GetViewFn)block->vtbl->slot[BFABI_SLOT_GetView];
      BfMat *blockView = getView ? getView((BfMat *)block) : NULL;
      void *viewStandIn = blockView ? NULL : xmalloc(64); /* the block view the reference mallocs here */
      if (getView && !blockView) setError(BFABI_ERROR_MEMORY_ERROR);
      if (blockView) block = blockView;
      if (bfMatGetNumRows(block) != i1 - i0 || bfMatGetNumCols(block) != j1 - j0) setError(BFABI_ERROR_INCOMPATIBLE_SHAPES);
      BfMat *tmp = (op2Rows && !currentError) ? bfMatMul(block, op2Rows) : NULL;
      if (tmp) bfMatAddInplace(resultRows, tmp);
      bfMatDelete(&tmp);
      bfMatDelete(&op2Rows);
      bfMatDelete(&blockView);
      free(viewStandIn);
      if (currentError) { bfMatDelete(&resultRows); bfMatDelete(&result); return NULL; }
    }
    bfMatDelete(&resultRows);
  }
  return result;
}
/* src/mat_block_dense.c:574-638 */
static BfVec *blockDenseMulVec(BfMat const *mat, BfVec const *vec) {
  BfAbiMatBlock const *matBlock = (BfAbiMatBlock const *)mat;
  if (((BfAbiVecGetTypeFn)vec->vtbl->slot[BFABI_VSLOT_GetType])(vec) != BFABI_TYPE_VEC_REAL) { setError(BFABI_ERROR_NOT_IMPLEMENTED); return NULL; }
  size_t numRowBlocks = mat->numRows, numColBlocks = mat->numCols;
  BfVec *result = vecRealNewWithValue(bfMatGetNumRows(mat), 0);
  for (size_t i = 0; i < numRowBlocks; ++i) {
    size_t i0 = matBlock->rowOffset[i], i1 = matBlock->rowOffset[i + 1];
    if (i0 == i1) continue;
    BfVec *resultSubvecView = vecRealGetSubvecView(result, i0, i1);
    for (size_t j = 0; j < numColBlocks; ++j) {
      BfMat const *block = matBlock->block[i * numColBlocks + j];
      size_t j0 = matBlock->colOffset[j], j1 = matBlock->colOffset[j + 1];
      if (j0 == j1) continue;
      BfVec *subvecView = vecRealGetSubvecView((BfVec *)vec, j0, j1);
      BfVec *tmp = bfMatMulVec(block, subvecView);
      if (tmp) vecRealAddInplace(resultSubvecView, tmp);
      bfVecDelete(&tmp);
      bfVecDelete(&subvecView);
      if (currentError) { bfVecDelete(&resultSubvecView); bfVecDelete(&result); return NULL; }
    }
    bfVecDelete(&resultSubvecView);
  }
  return result;
}
/* src/mat_block_dense.c:696-758 */
static BfVec *blockDenseRmulVec(BfMat const *mat, BfVec const *vec) {
  BfAbiMatBlock const *matBlock = (BfAbiMatBlock const *)mat;
  if (((BfAbiVecGetTypeFn)vec->vtbl->slot[BFABI_VSLOT_GetType])(vec) != BFABI_TYPE_VEC_REAL) { setError(BFABI_ERROR_NOT_IMPLEMENTED); return NULL; }
  size_t numRowBlocks = mat->numRows, numColBlocks = mat->numCols;
  BfVec *result = vecRealNewWithValue(bfMatGetNumCols(mat), 0);
  for (size_t j = 0; j < numColBlocks; ++j) {
    size_t j0 = matBlock->colOffset[j], j1 = matBlock->colOffset[j + 1];
    if (j0 == j1) continue;
    BfVec *resultSubvecView = vecRealGetSubvecView(result, j0, j1);
    for (size_t i = 0; i < numRowBlocks; ++i) {
      BfMat const *block = matBlock->block[i * numColBlocks + j];
      size_t i0 = matBlock->rowOffset[i], i1 = matBlock->rowOffset[i + 1];
      if (i0 == i1) continue;
      BfVec *subvecView = vecRealGetSubvecView((BfVec *)vec, i0, i1);
      BfVec *tmp = bfMatRmulVec(block, subvecView);
      if (tmp) vecRealAddInplace(resultSubvecView, tmp);
      bfVecDelete(&tmp);
      bfVecDelete(&subvecView);
      if (currentError) { bfVecDelete(&resultSubvecView); bfVecDelete(&result); return NULL; }
    }
    bfVecDelete(&resultSubvecView);
  }
  return result;
}
/* src/mat_block_dense.c:950-986: the matrix of block pointers transposed, extents and offsets swapped, every block
 * transposed in place.  (BfMatBlockCoo has no Transpose: its slot is NULL in the reference too.) */
static void blockDenseTranspose(BfMat *mat) {
  BfAbiMatBlock *b = (BfAbiMatBlock *)mat;
  size_t const nbr = mat->numRows, nbc = mat->numCols;
  BfMat **bt = xmalloc((nbr * nbc + 1) * sizeof(BfMat *));
  size_t k = 0;
  for (size_t j = 0; j < nbc; ++j)
    for (size_t i = 0; i < nbr; ++i) bt[k++] = b->block[i * nbc + j];
  mat->numRows = nbc; mat->numCols = nbr;
  size_t *o = b->rowOffset; b->rowOffset = b->colOffset; b->colOffset = o;
  free(b->block);
  b->block = bt;
  for (k = 0; k < nbr * nbc; ++k) {
    bfMatTranspose(b->block[k]);
    if (currentError) return;
  }
}
static BfAbiMatVtable MatBlockDenseVtable = {.slot = {
  [BFABI_SLOT_Transpose] = (void *)blockDenseTranspose,
  [BFABI_SLOT_RmulVec] = (void *)blockDenseRmulVec,
  [BFABI_SLOT_Delete] = (void *)blockDenseDelete,
  [BFABI_SLOT_GetType] = (void *)blockDenseGetType,
  [BFABI_SLOT_NumBytes] = (void *)blockDenseNumBytes,
  [BFABI_SLOT_GetNumRows] = (void *)blockGetNumRows,
  [BFABI_SLOT_GetNumCols] = (void *)blockGetNumCols,
  [BFABI_SLOT_Mul] = (void *)blockDenseMul,
  [BFABI_SLOT_MulVec] = (void *)blockDenseMulVec,
}};
BfMat *bfMatBlockDenseNewFromBlocks(size_t nbr, size_t nbc, size_t const *rowOffset, size_t const *colOffset, BfMat **blocks) {
  BfAbiMatBlock *b = xmalloc(sizeof *b);
  blockInitCommon(b, &MatBlockDenseVtable, nbr, nbc, nbr * nbc, rowOffset, colOffset, blocks);
  return &b->super;
}

/* =========================================================================
 * MatProduct: src/mat_product.c
 * ========================================================================= */
static BfAbiMatVtable MatProductVtable;
static int productGetType(BfMat const *m) { (void)m; return BFABI_TYPE_MAT_PRODUCT; }
static size_t productNumFactors(BfMat const *m) { return ((BfAbiMatProduct const *)m)->factorArr.num_elts; }
static BfMat *productFactor(BfMat const *m, size_t i) { return (BfMat *)((BfAbiMatProduct const *)m)->factorArr.data[i]; }
/* :146-192: rows from first factor, cols from last */
static size_t productGetNumRows(BfMat const *m) { return bfMatGetNumRows(productFactor(m, 0)); }
static size_t productGetNumCols(BfMat const *m) { return bfMatGetNumCols(productFactor(m, productNumFactors(m) - 1)); }
static size_t productNumBytes(BfMat const *m) {
  size_t n = 0;
  for (size_t i = 0; i < productNumFactors(m); ++i) n += bfMatNumBytes(productFactor(m, i));
  return n;
}
static void productDelete(BfMat **mat) {
  BfAbiMatProduct *p = (BfAbiMatProduct *)*mat;
  for (size_t i = 0; i < p->factorArr.num_elts; ++i) { BfMat *f = p->factorArr.data[i]; bfMatDelete(&f); }
  free(p->factorArr.data);
  free(p);
  *mat = NULL;
}
/* :211-245 */
static BfMat *productMul(BfMat const *matProduct, BfMat const *otherMat) {
  size_t numFactors = productNumFactors(matProduct);
  size_t i = numFactors - 1;
  BfMat *prev = bfMatMul(productFactor(matProduct, i), otherMat);
  BfMat *result = prev;
  if (currentError) { bfMatDelete(&prev); return NULL; }
  while (i > 0) {
    result = bfMatMul(productFactor(matProduct, --i), prev);
    bfMatDelete(&prev);
    if (currentError) { bfMatDelete(&result); return NULL; }
    prev = result;
  }
  return result;
}
/* :282-310 Rmul: mat * F0 * F1 * ... * F_{L-1}, factors in order */
static BfMat *productRmul(BfMat const *matProduct, BfMat const *mat) {
  if (bfMatGetNumCols(mat) != bfMatGetNumRows(matProduct)) { setError(BFABI_ERROR_INVALID_ARGUMENTS); return NULL; }
  BfMat *prev = bfMatRmul(productFactor(matProduct, 0), mat);
  BfMat *result = prev;
  if (currentError) { bfMatDelete(&prev); return NULL; }
  for (size_t i = 1; i < productNumFactors(matProduct); ++i) {
    result = bfMatRmul(productFactor(matProduct, i), prev);
    bfMatDelete(&prev);
    if (currentError) { bfMatDelete(&result); return NULL; }
    prev = result;
  }
  return result;
}
/* :247-280 */
static BfVec *productMulVec(BfMat const *matProduct, BfVec const *vec) {
  size_t numFactors = productNumFactors(matProduct);
  size_t i = numFactors - 1;
  BfVec *prev = bfMatMulVec(productFactor(matProduct, i), vec);
  BfVec *result = prev;
  if (currentError) { bfVecDelete(&prev); return NULL; }
  while (i > 0) {
    result = bfMatMulVec(productFactor(matProduct, --i), prev);
    bfVecDelete(&prev);
    if (currentError) { bfVecDelete(&result); return NULL; }
    prev = result;
  }
  return result;
}
/* src/mat_product.c:314-345: factors in order 0, 1, ..., L-1 */
static BfVec *productRmulVec(BfMat const *matProduct, BfVec const *vec) {
  size_t numFactors = productNumFactors(matProduct);
  size_t i = 0;
  BfVec *prev = bfMatRmulVec(productFactor(matProduct, i), vec);
  BfVec *result = prev;
  if (currentError) { bfVecDelete(&prev); return NULL; }
  while (++i < numFactors) {
    result = bfMatRmulVec(productFactor(matProduct, i), prev);
    bfVecDelete(&prev);
    if (currentError) { bfVecDelete(&result); return NULL; }
    prev = result;
  }
  return result;
}
/* :409-420: the factors reversed, each transposed in place */
static void productTranspose(BfMat *mat) {
  BfAbiMatProduct *p = (BfAbiMatProduct *)mat;
  size_t const nf = p->factorArr.num_elts;
  for (size_t i = 0; i < nf / 2; ++i) { void *t = p->factorArr.data[i]; p->factorArr.data[i] = p->factorArr.data[nf - 1 - i]; p->factorArr.data[nf - 1 - i] = t; }
  for (size_t i = 0; i < nf; ++i) {
    bfMatTranspose(p->factorArr.data[i]);
    if (currentError) return;
  }
}
static BfAbiMatVtable MatProductVtable = {.slot = {
  [BFABI_SLOT_Transpose] = (void *)productTranspose,
  [BFABI_SLOT_RmulVec] = (void *)productRmulVec,
  [BFABI_SLOT_Delete] = (void *)productDelete,
  [BFABI_SLOT_GetType] = (void *)productGetType,
  [BFABI_SLOT_NumBytes] = (void *)productNumBytes,
  [BFABI_SLOT_GetNumRows] = (void *)productGetNumRows,
  [BFABI_SLOT_GetNumCols] = (void *)productGetNumCols,
  [BFABI_SLOT_Mul] = (void *)productMul,
  [BFABI_SLOT_Rmul] = (void *)productRmul,
  [BFABI_SLOT_MulVec] = (void *)productMulVec,
}};
BfMat *bfMatProductNewFromFactors(size_t numFactors, BfMat **factors) {
  BfAbiMatProduct *p = xmalloc(sizeof *p);
  p->super.vtbl = &MatProductVtable;
  p->super.props = 0;
  p->super.numRows = p->super.numCols = (size_t)-1; /* BF_SIZE_BAD_VALUE, as bfMatProductInit */
  p->factorArr.data = xmalloc(numFactors * sizeof(void *));
  memcpy(p->factorArr.data, factors, numFactors * sizeof(void *));
  p->factorArr.capacity = numFactors;
  p->factorArr.num_elts = numFactors;
  p->factorArr.isView = false;
  return &p->super;
}

/* =========================================================================
 * MatSum / MatCooComplex / MatDiagReal: the decorations bfMatBlockDenseAddInplace
 * leaves in a system matrix (src/mat_block_dense.c:458-510)
 * ========================================================================= */
static BfAbiMatVtable MatSumVtable;
static int sumGetType(BfMat const *m) { (void)m; return BFABI_TYPE_MAT_SUM; }
static BfMat *sumTerm(BfMat const *m, size_t i) { return (BfMat *)((BfAbiMatSum const *)m)->termArr.data[i]; }
static size_t sumNumTerms(BfMat const *m) { return ((BfAbiMatSum const *)m)->termArr.num_elts; }
static size_t sumGetNumRows(BfMat const *m) { return bfMatGetNumRows(sumTerm(m, 0)); }
static size_t sumGetNumCols(BfMat const *m) { return bfMatGetNumCols(sumTerm(m, 0)); }
static size_t sumNumBytes(BfMat const *m) { size_t n = 0; for (size_t i = 0; i < sumNumTerms(m); ++i) n += bfMatNumBytes(sumTerm(m, i)); return n; }
static void sumDelete(BfMat **mat) {
  BfAbiMatSum *s = (BfAbiMatSum *)*mat;
  for (size_t i = 0; i < s->termArr.num_elts; ++i) { BfMat *t = s->termArr.data[i]; bfMatDelete(&t); }
  free(s->termArr.data);
  free(s);
  *mat = NULL;
}
/* src/mat_sum.c:54-83 */
static BfMat *sumMul(BfMat const *mat, BfMat const *otherMat) {
  size_t m = bfMatGetNumRows(mat), p = bfMatGetNumCols(mat), n = bfMatGetNumCols(otherMat);
  if (p != bfMatGetNumRows(otherMat)) { setError(BFABI_ERROR_INVALID_ARGUMENTS); return NULL; }
  BfMat *result = bfMatZerosLike(otherMat, m, n);
  for (size_t i = 0; i < sumNumTerms(mat); ++i) {
    BfMat *prod = bfMatMul(sumTerm(mat, i), otherMat);
    if (prod) bfMatAddInplace(result, prod);
    bfMatDelete(&prod);
    if (currentError) { bfMatDelete(&result); return NULL; }
  }
  return result;
}
static BfAbiMatVtable MatSumVtable = {.slot = {
  [BFABI_SLOT_Delete] = (void *)sumDelete,
  [BFABI_SLOT_GetType] = (void *)sumGetType,
  [BFABI_SLOT_NumBytes] = (void *)sumNumBytes,
  [BFABI_SLOT_GetNumRows] = (void *)sumGetNumRows,
  [BFABI_SLOT_GetNumCols] = (void *)sumGetNumCols,
  [BFABI_SLOT_Mul] = (void *)sumMul,
}};
BfMat *bfMatSumNewFromTerms(size_t numTerms, BfMat **terms) {
  BfAbiMatSum *s = xmalloc(sizeof *s);
  s->super.vtbl = &MatSumVtable;
  s->super.props = 0;
  s->super.numRows = s->super.numCols = (size_t)-1;
  s->termArr.data = xmalloc(numTerms * sizeof(void *));
  memcpy(s->termArr.data, terms, numTerms * sizeof(void *));
  s->termArr.capacity = s->termArr.num_elts = numTerms;
  s->termArr.isView = false;
  return &s->super;
}

static BfAbiMatVtable MatCooComplexVtable;
static int cooComplexAssignQuirk = 0;
void bfrefCooComplexAssignQuirk(int on) { cooComplexAssignQuirk = on; }
static int cooComplexGetType(BfMat const *m) { (void)m; return BFABI_TYPE_MAT_COO_COMPLEX; }
static size_t cooComplexNumBytes(BfMat const *m) { return ((BfAbiMatCooComplex const *)m)->numElts * (2 * sizeof(size_t) + sizeof(cplx)); }
static void cooComplexDelete(BfMat **mat) {
  BfAbiMatCooComplex *c = (BfAbiMatCooComplex *)*mat;
  free(c->rowInd); free(c->colInd); free(c->value);
  free(c);
  *mat = NULL;
}
/* mul_denseComplex, src/mat_coo_complex.c:212-262.  The reference writes
 * `*outPtr = *inPtr; *outPtr *= z;` (:248-251): the product of an entry
 * REPLACES the result row, so the last stored entry of a row wins.  Default here
 * is the evident intent (accumulate); bfrefCooComplexAssignQuirk(1) restates the
 * code as written. */
static BfMat *cooComplexMul(BfMat const *mat, BfMat const *otherMat) {
  if (bfMatGetType(otherMat) != BFABI_TYPE_MAT_DENSE_COMPLEX) { setError(BFABI_ERROR_NOT_IMPLEMENTED); return NULL; }
  BfAbiMatCooComplex const *c = (BfAbiMatCooComplex const *)mat;
  BfAbiMatDenseComplex const *x = (BfAbiMatDenseComplex const *)otherMat;
  size_t m = mat->numRows, n = mat->numCols, p = otherMat->numCols;
  if (n != otherMat->numRows) { setError(BFABI_ERROR_INVALID_ARGUMENTS); return NULL; }
  BfMat *result = bfMatZerosLike(otherMat, m, p);
  BfAbiMatDenseComplex *y = (BfAbiMatDenseComplex *)result;
  cplx *Y = (cplx *)y->data;
  cplx const *X = (cplx const *)x->data;
  for (size_t k = 0; k < c->numElts; ++k) {
    size_t i = c->rowInd[k], j = c->colInd[k];
    cplx z = ((cplx const *)c->value)[k];
    for (size_t l = 0; l < p; ++l) {
      cplx v = X[j * x->rowStride + l * x->colStride] * z;
      if (cooComplexAssignQuirk) Y[i * y->rowStride + l] = v; else Y[i * y->rowStride + l] += v;
    }
  }
  return result;
}
static BfAbiMatVtable MatCooComplexVtable = {.slot = {
  [BFABI_SLOT_Delete] = (void *)cooComplexDelete,
  [BFABI_SLOT_GetType] = (void *)cooComplexGetType,
  [BFABI_SLOT_NumBytes] = (void *)cooComplexNumBytes,
  [BFABI_SLOT_GetNumRows] = (void *)plainGetNumRows,
  [BFABI_SLOT_GetNumCols] = (void *)plainGetNumCols,
  [BFABI_SLOT_Mul] = (void *)cooComplexMul,
}};
BfMat *bfMatCooComplexNewFromArrays(size_t m, size_t n, size_t numElts, size_t const *rowInd, size_t const *colInd, double const *value) {
  BfAbiMatCooComplex *c = xmalloc(sizeof *c);
  c->super.vtbl = &MatCooComplexVtable;
  c->super.props = 0;
  c->super.numRows = m; c->super.numCols = n;
  c->numElts = c->capacity = numElts;
  c->rowInd = xmalloc(numElts * sizeof(size_t)); memcpy(c->rowInd, rowInd, numElts * sizeof(size_t));
  c->colInd = xmalloc(numElts * sizeof(size_t)); memcpy(c->colInd, colInd, numElts * sizeof(size_t));
  c->value = xmalloc(numElts * sizeof(cplx)); memcpy(c->value, value, numElts * sizeof(cplx));
  return &c->super;
}

static BfAbiMatVtable MatDiagRealVtable;
static int diagRealGetType(BfMat const *m) { (void)m; return BFABI_TYPE_MAT_DIAG_REAL; }
static size_t diagRealNumBytes(BfMat const *m) { return ((BfAbiMatDiagReal const *)m)->numElts * sizeof(double); }
static void diagRealDelete(BfMat **mat) { BfAbiMatDiagReal *d = (BfAbiMatDiagReal *)*mat; free(d->data); free(d); *mat = NULL; }
/* bfMatDiagRealMulVec, src/mat_diag_real.c:168-: y = d .* x (real vectors) */
static BfVec *diagRealMulVec(BfMat const *mat, BfVec const *vec) {
  BfAbiMatDiagReal const *d = (BfAbiMatDiagReal const *)mat;
  if (mat->numCols != vec->size) { setError(BFABI_ERROR_INVALID_ARGUMENTS); return NULL; }
  BfAbiVecReal const *x = (BfAbiVecReal const *)vec;
  BfVec *res = vecRealNewWithValue(mat->numRows, 0);
  double *y = ((BfAbiVecReal *)res)->data;
  for (size_t i = 0; i < d->numElts; ++i) y[i] = d->data[i] * x->data[i * x->stride];
  return res;
}
static BfAbiMatVtable MatDiagRealVtable = {.slot = {
  [BFABI_SLOT_Delete] = (void *)diagRealDelete,
  [BFABI_SLOT_GetType] = (void *)diagRealGetType,
  [BFABI_SLOT_NumBytes] = (void *)diagRealNumBytes,
  [BFABI_SLOT_GetNumRows] = (void *)plainGetNumRows,
  [BFABI_SLOT_GetNumCols] = (void *)plainGetNumCols,
  [BFABI_SLOT_MulVec] = (void *)diagRealMulVec,
  [BFABI_SLOT_RmulVec] = (void *)diagRealMulVec,
}};
BfMat *bfMatDiagRealNewFromPtr(size_t m, size_t n, size_t numElts, double const *data) {
  BfAbiMatDiagReal *d = xmalloc(sizeof *d);
  d->super.vtbl = &MatDiagRealVtable;
  d->super.props = 0;
  d->super.numRows = m; d->super.numCols = n;
  d->numElts = numElts;
  d->data = xmalloc(numElts * sizeof(double)); memcpy(d->data, data, numElts * sizeof(double));
  return &d->super;
}

/* =========================================================================
 * Graph from a flat descriptor
 * ========================================================================= */
static int cmp_size(void const *a, void const *b) {
  size_t x = *(size_t const *)a, y = *(size_t const *)b;
  return x < y ? -1 : x > y;
}
static size_t uniq(size_t *a, size_t n) {
  if (!n) return 0;
  size_t w = 1;
  for (size_t i = 1; i < n; ++i) if (a[i] != a[w - 1]) a[w++] = a[i];
  return w;
}
static size_t findIndex(size_t const *a, size_t n, size_t v) {
  size_t lo = 0, hi = n;
  while (lo < hi) { size_t mid = (lo + hi) / 2; if (a[mid] < v) lo = mid + 1; else hi = mid; }
  return lo;
}

static int fromDescTyped = 0;        /* set by bfrefMatFromDescTyped for the duration of one build */
BfMat *bfrefMatFromDesc(BfhipDesc const *d, uint64_t seed, uint64_t rootOverride);
static BfMat *fromDescRec(BfhipDesc const *d, uint64_t const *bases, uint64_t seed, uint64_t node) {
  uint64_t m = d->rows[node], n = d->cols[node];
  switch (d->kind[node]) {
  case BFHIP_NODE_DENSE: {
    void const *src = d->leafData ? d->leafData[node] : NULL;
    uint64_t ld = (d->leafRowStride && src) ? d->leafRowStride[node] : n;
    if (d->dtype == BFHIP_C128) {
      double *buf = xmalloc(m * n * 2 * sizeof(double));
      if (src) {
        for (uint64_t i = 0; i < m; ++i) memcpy(buf + 2 * i * n, (double const *)src + 2 * i * ld, 2 * n * sizeof(double));
      } else {
        double scale = sqrt(3.0 / (2.0 * (double)n));
        for (uint64_t e = 0; e < m * n; ++e) {
          buf[2 * e] = bfhip_synth_value(seed, bases[node] + e, 0) * scale;
          buf[2 * e + 1] = bfhip_synth_value(seed, bases[node] + e, 1) * scale;
        }
      }
      return bfMatDenseComplexNewFromPtr(m, n, buf, 2);
    } else {
      double *buf = xmalloc(m * n * sizeof(double));
      if (src) {
        for (uint64_t i = 0; i < m; ++i) memcpy(buf + i * n, (double const *)src + i * ld, n * sizeof(double));
      } else {
        double scale = sqrt(3.0 / (double)n);
        for (uint64_t e = 0; e < m * n; ++e) buf[e] = bfhip_synth_value(seed, bases[node] + e, 0) * scale;
      }
      return bfMatDenseRealNewFromPtr(m, n, buf, 2);
    }
  }
  case BFHIP_NODE_IDENTITY:
    return bfMatIdentityNew(m);
  case BFHIP_NODE_PRODUCT: {
    uint64_t b = d->childBegin[node], e = d->childBegin[node + 1];
    BfMat **f = xmalloc((e - b) * sizeof(BfMat *));
    for (uint64_t c = b; c < e; ++c) f[c - b] = fromDescRec(d, bases, seed, d->childNode[c]);
    BfMat *p = bfMatProductNewFromFactors(e - b, f);
    free(f);
    return p;
  }
  case BFHIP_NODE_BLOCK: {
    uint64_t b = d->childBegin[node], e = d->childBegin[node + 1], nc = e - b;
    size_t *rb = xmalloc((2 * nc + 2) * sizeof(size_t)), *cb = xmalloc((2 * nc + 2) * sizeof(size_t));
    size_t nr = 0, ncb = 0;
    rb[nr++] = 0; rb[nr++] = m; cb[ncb++] = 0; cb[ncb++] = n;
    for (uint64_t c = b; c < e; ++c) {
      uint64_t ch = d->childNode[c];
      rb[nr++] = d->childRow0[c]; rb[nr++] = d->childRow0[c] + d->rows[ch];
      cb[ncb++] = d->childCol0[c]; cb[ncb++] = d->childCol0[c] + d->cols[ch];
    }
    qsort(rb, nr, sizeof(size_t), cmp_size); nr = uniq(rb, nr);
    qsort(cb, ncb, sizeof(size_t), cmp_size); ncb = uniq(cb, ncb);
    BfMat **blocks = xmalloc(nc * sizeof(BfMat *));
    size_t *ri = xmalloc(nc * sizeof(size_t)), *ci = xmalloc(nc * sizeof(size_t));
    for (uint64_t c = b; c < e; ++c) {
      uint64_t ch = d->childNode[c];
      size_t i = findIndex(rb, nr, d->childRow0[c]), j = findIndex(cb, ncb, d->childCol0[c]);
      /* each child must span exactly one block row and one block column */
      if (rb[i + 1] != d->childRow0[c] + d->rows[ch] || cb[j + 1] != d->childCol0[c] + d->cols[ch]) setError(BFABI_ERROR_INCOMPATIBLE_SHAPES);
      ri[c - b] = i; ci[c - b] = j;
      blocks[c - b] = fromDescRec(d, bases, seed, ch);
    }
    BfMat *r = NULL;
    /* bfrefMatFromDescTyped: a BLOCK node the descriptor calls BlockDiag / BlockDense becomes that container when its
     * children are exactly the diagonal / the full grid (else, and by default, the general BlockCoo) */
    uint8_t const want = (fromDescTyped && d->blockKind) ? d->blockKind[node] : 0;
    if (want == BFABI_TYPE_MAT_BLOCK_DIAG && nc && nr - 1 == nc && ncb - 1 == nc) {
      BfMat **ordered = xmalloc(nc * sizeof(BfMat *));
      int ok = 1;
      for (size_t k = 0; k < nc; ++k) ordered[k] = NULL;
      for (size_t k = 0; k < nc; ++k) { if (ri[k] != ci[k] || ordered[ri[k]]) ok = 0; else ordered[ri[k]] = blocks[k]; }
      if (ok) r = bfMatBlockDiagNewFromBlocks(nc, ordered);
      free(ordered);
    } else if (want == BFABI_TYPE_MAT_BLOCK_DENSE && nc && (nr - 1) * (ncb - 1) == nc) {
      BfMat **grid = xmalloc(nc * sizeof(BfMat *));
      int ok = 1;
      for (size_t k = 0; k < nc; ++k) grid[k] = NULL;
      for (size_t k = 0; k < nc; ++k) { size_t const at = ri[k] * (ncb - 1) + ci[k]; if (grid[at]) ok = 0; else grid[at] = blocks[k]; }
      if (ok) r = bfMatBlockDenseNewFromBlocks(nr - 1, ncb - 1, rb, cb, grid);
      free(grid);
    }
    if (!r) r = bfMatBlockCooNewFromArrays(nr - 1, ncb - 1, nc, rb, cb, ri, ci, blocks);
    free(rb); free(cb); free(blocks); free(ri); free(ci);
    return r;
  }
  }
  setError(BFABI_ERROR_TYPE_ERROR);
  return NULL;
}

BfMat *bfrefMatFromDescTyped(BfhipDesc const *d, uint64_t seed, uint64_t rootOverride) {
  fromDescTyped = 1;
  BfMat *r = bfrefMatFromDesc(d, seed, rootOverride);
  fromDescTyped = 0;
  return r;
}

BfMat *bfrefMatFromDesc(BfhipDesc const *d, uint64_t seed, uint64_t rootOverride) {
  uint64_t *bases = xmalloc((d->numNodes + 1) * sizeof(uint64_t));
  uint64_t acc = 0;
  for (uint64_t i = 0; i < d->numNodes; ++i) {
    bases[i] = acc;
    if (d->kind[i] == BFHIP_NODE_DENSE) acc += d->rows[i] * d->cols[i];
  }
  BfMat *r = fromDescRec(d, bases, seed, rootOverride == UINT64_MAX ? d->root : rootOverride);
  free(bases);
  if (currentError) { bfMatDelete(&r); return NULL; }
  return r;
}
