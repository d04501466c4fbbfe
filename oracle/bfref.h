/* bfref.h -- ORACLE (test infrastructure, not product code).
 *
 * A plain-C restatement of the reference's apply path: a miniature BfMat
 * object system whose structs are byte-compatible with the reference's
 * (layouts from include/bfhip_abi.h) and whose Mul/MulVec follow the
 * reference's algorithms statement by statement (same block order, same
 * allocate-view / multiply / accumulate sequence).  Each function cites the
 * reference file:line it follows.
 *
 * PARITY STATUS: the reference's tests hold no fixture for this path (SURVEY.md
 * section 4) and the reference itself cannot be built in this image without
 * stand-in BLAS/LAPACKE headers (include/bf/blas.h:3-9 needs <openblas/cblas.h>,
 * <lapacke.h>; absent), so there is no oracle/_ref: "parity unpinned" by
 * reference test fixtures.  The restatement is pinned instead by outputs of the
 * REAL reference that the survey recorded: the
 * ||A_BF x||^2 checksums of four configurations (N = 4096 ... 65536), reproduced
 * to 13-15 digits (tests/test_oracle_golden.py on the CPU for N = 4096,
 * tests/test_gpu_build.py for all four), the exact block-layout statistics of
 * two configurations (tests/test_structure.py), and analytic known answers
 * (dense Helmholtz kernel matvecs, the reference examples' own acceptance check
 * examples/simple/bf_all_blocks.c:149-153).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * use this library.  The shipped engine (libbfhip.so) never links or loads it.
 */
#ifndef BFREF_H
#define BFREF_H

#include <stddef.h>
#include <stdint.h>
#include "../include/bfhip_abi.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef BfAbiMat BfMat;
typedef BfAbiVec BfVec;

/* ---- public interface, reference names (include/bf/mat.h:43-110) ---------- */
BfMat *bfMatMul(BfMat const *lhs, BfMat const *rhs);          /* src/mat.c:183 */
BfVec *bfMatMulVec(BfMat const *lhs, BfVec const *rhs);       /* src/mat.c:187 */
BfVec *bfMatRmulVec(BfMat const *lhs, BfVec const *rhs);      /* src/mat.c:197: x^T A as a vector */
void bfMatDelete(BfMat **mat);                                /* src/mat.c:43  */
size_t bfMatGetNumRows(BfMat const *mat);                     /* src/mat.c:79  */
size_t bfMatGetNumCols(BfMat const *mat);                     /* src/mat.c:83  */
int bfMatGetType(BfMat const *mat);                           /* src/mat.c:55  */
size_t bfMatNumBytes(BfMat const *mat);                       /* src/mat.c:59  */
void bfVecDelete(BfVec **vec);

/* ---- constructors --------------------------------------------------------- */
/* policy: 0 = copy `data`, 1 = view (caller keeps ownership), 2 = steal (free()d on delete) */
BfMat *bfMatDenseComplexNewFromPtr(size_t m, size_t n, double *data, int policy);
BfMat *bfMatDenseComplexNewZeros(size_t m, size_t n);
BfMat *bfMatDenseRealNewFromPtr(size_t m, size_t n, double *data, int policy);
BfMat *bfMatIdentityNew(size_t n);
/* containers steal their children and copy the index arrays */
BfMat *bfMatBlockDiagNewFromBlocks(size_t numBlocks, BfMat **blocks);
BfMat *bfMatBlockCooNewFromArrays(size_t numBlockRows, size_t numBlockCols, size_t numBlocks,
                                  size_t const *rowOffset, size_t const *colOffset,
                                  size_t const *rowInd, size_t const *colInd, BfMat **blocks);
BfMat *bfMatBlockDenseNewFromBlocks(size_t numBlockRows, size_t numBlockCols,
                                    size_t const *rowOffset, size_t const *colOffset, BfMat **blocks);
BfMat *bfMatProductNewFromFactors(size_t numFactors, BfMat **factors);
/* decorations of a system matrix (mat_sum.h, mat_coo_complex.h, mat_diag_real.h) */
BfMat *bfMatSumNewFromTerms(size_t numTerms, BfMat **terms);
BfMat *bfMatCooComplexNewFromArrays(size_t m, size_t n, size_t numElts, size_t const *rowInd, size_t const *colInd, double const *value);
BfMat *bfMatDiagRealNewFromPtr(size_t m, size_t n, size_t numElts, double const *data);
/* 1: bfMatCooComplexMul assigns instead of accumulating, as the reference's code does (src/mat_coo_complex.c:248-251) */
void bfrefCooComplexAssignQuirk(int on);

BfVec *bfVecRealNewFromPtr(size_t n, double *data, int policy);
double *bfVecRealData(BfVec *vec);
double *bfMatDenseData(BfMat *mat);   /* complex or real dense */

/* global error code, as src/error.c:7-24 but non-fatal */
int bfGetError(void);
void bfClearError(void);
void bfSetError(int error);             /* src/error.c:20-24 (non-fatal here) */

/* counters for the cpu_baseline report */
typedef struct BfrefCounters {
  uint64_t gemmCalls;   /* zgemm/dgemv-equivalent calls */
  uint64_t macs;        /* multiply-accumulates (complex or real) */
  uint64_t mallocs;     /* heap allocations made by the apply path */
} BfrefCounters;
void bfrefResetCounters(void);
void bfrefGetCounters(BfrefCounters *out);

/* optional BLAS backend: dlopen a CBLAS (e.g. the OpenBLAS bundled with
 * numpy/scipy wheels) and route leaf products through its zgemm/dgemv.
 * symbolPrefix is prepended to "cblas_zgemm"/"cblas_dgemv" ("scipy_").
 * Returns 0 on success; on failure the built-in C kernels stay active. */
int bfrefUseBlas(char const *path, char const *symbolPrefix);
char const *bfrefBlasName(void);

/* ---- graph from a flat descriptor (include/bfhip.h: BfhipDesc) ------------ */
struct BfhipDesc;
/* Build the BfMat graph a descriptor describes.  BLOCK nodes become
 * BfMatBlockCoo whose block rows/cols are the distinct row/col offsets of the
 * children; PRODUCT nodes BfMatProduct; leaves with NULL data are filled
 * with the engine's synthetic value stream for `seed` (same values the
 * device generates).  `rootOverride` == UINT64_MAX uses desc->root. */
BfMat *bfrefMatFromDesc(struct BfhipDesc const *desc, uint64_t seed, uint64_t rootOverride);
/* the same, but BLOCK nodes the descriptor calls BlockDiag / BlockDense (blockKind) become those containers when their
 * children are exactly the diagonal / the full grid: the graphs bfMatTranspose can walk (BlockCoo has no Transpose slot) */
BfMat *bfrefMatFromDescTyped(struct BfhipDesc const *desc, uint64_t seed, uint64_t rootOverride);

#ifdef __cplusplus
}
#endif
#endif
