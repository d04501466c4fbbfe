/* bfhip_abi.h -- byte-compatible views of the reference's BfMat object graph.
 *
 * The engine is a drop-in behind the reference's `BfMat` vtable, so it has to
 * read (never write) object graphs built by `fac_helm2` / `fac_streamer`.
 * This header re-declares, under `BfAbi*` names that cannot clash with the
 * reference's own headers, exactly the fields the apply path touches:
 *
 *   struct BfMat            reference include/bf/mat.h:187-196
 *   struct BfMatVtable      reference include/bf/mat.h:112-179  (66 slots)
 *   enum   BfMatProps       reference include/bf/mat.h:30-39
 *   enum   BfType           reference include/bf/types.h:78-144 (positional)
 *   struct BfPtrArray       reference include/bf/ptr_array.h:7-12
 *   struct BfMatProduct     reference include/bf/mat_product.h:28-31
 *   struct BfMatBlock       reference include/bf/mat_block.h:39-66
 *   struct BfMatBlockCoo    reference include/bf/mat_block_coo.h:34-46
 *   struct BfMatDenseComplex reference include/bf/mat_dense_complex.h:69-81
 *   struct BfMatDense/Real  reference include/bf/mat_dense.h:15-20,
 *                                     include/bf/mat_dense_real.h:54-57
 *   struct BfMatIdentity    reference include/bf/mat_identity.h:34-36
 *   struct BfVec/Real/Complex reference include/bf/vec.h:74-78,
 *                                     include/bf/vec_real.h:27-31
 *
 * Layout = x86-64 LP64, release build of the reference (BF_DEBUG undefined:
 * a debug build appends two pointers to BfMat, mat.h:192-195, and is refused
 * by bfhipCompile's sanity walk).  The vtable is declared as an array of 66
 * untyped slots plus the indices the engine needs; sizes and offsets are
 * pinned by _Static_asserts below and re-validated against the reference's
 * real headers by tests/test_abi_layout.py when /root/reference is present.
 */
#ifndef BFHIP_ABI_H
#define BFHIP_ABI_H

#include <stddef.h>
#include <stdint.h>
#include <stdbool.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- BfType numeric values (positional enum, types.h:78-144) ------------ */
enum {
  BFABI_TYPE_MAT = 0,
  BFABI_TYPE_MAT_COO_COMPLEX = 1,
  BFABI_TYPE_MAT_COO_REAL = 2,
  BFABI_TYPE_MAT_CSR_REAL = 3,
  BFABI_TYPE_MAT_DIAG_REAL = 4,
  BFABI_TYPE_MAT_DIFF = 5,
  BFABI_TYPE_MAT_FUNC = 6,
  BFABI_TYPE_MAT_GIVENS_COMPLEX = 7,
  BFABI_TYPE_MAT_IDENTITY = 8,
  BFABI_TYPE_MAT_PERM = 9,
  BFABI_TYPE_MAT_PRODUCT = 10,
  BFABI_TYPE_MAT_PYTHON = 11,
  BFABI_TYPE_MAT_SUM = 12,
  BFABI_TYPE_MAT_ZERO = 13,
  BFABI_TYPE_MAT_BLOCK = 14,
  BFABI_TYPE_MAT_BLOCK_COO = 15,
  BFABI_TYPE_MAT_BLOCK_DENSE = 16,
  BFABI_TYPE_MAT_BLOCK_DIAG = 17,
  BFABI_TYPE_MAT_DENSE = 18,
  BFABI_TYPE_MAT_DENSE_COMPLEX = 19,
  BFABI_TYPE_MAT_DENSE_REAL = 20,
  BFABI_TYPE_VEC = 26,
  BFABI_TYPE_VEC_COMPLEX = 27,
  BFABI_TYPE_VEC_REAL = 28,
  BFABI_TYPE_VEC_ZERO = 29
};

/* ---- BfMatProps bits (mat.h:30-39) --------------------------------------- */
enum {
  BFABI_MAT_PROPS_NONE = 0,
  BFABI_MAT_PROPS_VIEW = 1 << 0,
  BFABI_MAT_PROPS_TRANS = 1 << 1,
  BFABI_MAT_PROPS_CONJ = 1 << 2
};

/* ---- BfVecProps bits (vec.h:10-13) ---------------------------------------- */
enum {
  BFABI_VEC_PROPS_NONE = 0,
  BFABI_VEC_PROPS_VIEW = 1 << 0
};

/* ---- BfError values (error.h:3-16) ---------------------------------------- */
enum {
  BFABI_ERROR_NONE = 0,
  BFABI_ERROR_INVALID_ARGUMENTS = 1,
  BFABI_ERROR_RUNTIME_ERROR = 2,
  BFABI_ERROR_NOT_IMPLEMENTED = 3,
  BFABI_ERROR_MEMORY_ERROR = 4,
  BFABI_ERROR_OUT_OF_RANGE = 5,
  BFABI_ERROR_FILE_ERROR = 6,
  BFABI_ERROR_TYPE_ERROR = 7,
  BFABI_ERROR_INCOMPATIBLE_SHAPES = 8
};

/* ---- BfMatVtable slot indices (declaration order of mat.h:113-178) ------- */
enum {
  BFABI_SLOT_GetView = 0,
  BFABI_SLOT_Copy = 1,
  BFABI_SLOT_Steal = 2,
  BFABI_SLOT_Delete = 7,
  BFABI_SLOT_EmptyLike = 8,
  BFABI_SLOT_ZerosLike = 9,
  BFABI_SLOT_GetType = 10,
  BFABI_SLOT_NumBytes = 11,
  BFABI_SLOT_GetNumRows = 15,
  BFABI_SLOT_GetNumCols = 16,
  BFABI_SLOT_GetRowRange = 20,
  BFABI_SLOT_SetRowRange = 25,
  BFABI_SLOT_AddInplace = 37,
  BFABI_SLOT_Mul = 41,
  BFABI_SLOT_MulVec = 42,
  BFABI_SLOT_MulInplace = 43,
  BFABI_SLOT_Rmul = 44,
  BFABI_SLOT_RmulVec = 45,
  BFABI_SLOT_Transpose = 63,
  BFABI_NUM_MAT_SLOTS = 66
};

/* ---- BfVecVtable slot indices (vec.h:44-70) ------------------------------ */
enum {
  BFABI_VSLOT_Copy = 0,
  BFABI_VSLOT_Delete = 1,
  BFABI_VSLOT_GetType = 2,
  BFABI_VSLOT_GetSubvecCopy = 5,
  BFABI_VSLOT_GetSubvecView = 6,
  BFABI_VSLOT_GetSubvecViewConst = 7,
  BFABI_VSLOT_SetRange = 8,
  BFABI_VSLOT_AddInplace = 15,
  BFABI_NUM_VEC_SLOTS = 25
};

typedef struct BfAbiMatVtable {
  void *slot[BFABI_NUM_MAT_SLOTS];
} BfAbiMatVtable;

typedef struct BfAbiVecVtable {
  void *slot[BFABI_NUM_VEC_SLOTS];
} BfAbiVecVtable;

/* mat.h:187-196.  For block types numRows/numCols hold *block counts*
 * (mat_block.c:104, mat_block_coo.c:16-17), not element counts. */
typedef struct BfAbiMat {
  BfAbiMatVtable *vtbl;
  int props;        /* enum BfMatProps, 4 bytes + 4 pad */
  size_t numRows;
  size_t numCols;
} BfAbiMat;

typedef struct BfAbiPtrArray {
  void **data;
  size_t capacity;
  size_t num_elts;
  bool isView;
} BfAbiPtrArray;

typedef struct BfAbiMatProduct {
  BfAbiMat super;
  BfAbiPtrArray factorArr;   /* BfMat* factors; applied last-to-first */
} BfAbiMatProduct;

typedef struct BfAbiMatBlock {
  BfAbiMat super;
  void *vtbl;                /* BfMatBlockVtable* (12 slots), unused here */
  BfAbiMat **block;
  size_t *rowOffset;         /* numBlockRows+1 prefix sums */
  size_t *colOffset;         /* numBlockCols+1 prefix sums */
} BfAbiMatBlock;

typedef struct BfAbiMatBlockCoo {
  BfAbiMatBlock super;
  size_t numBlocks;
  size_t *rowInd;
  size_t *colInd;
} BfAbiMatBlockCoo;

/* BlockDiag / BlockDense add no fields (mat_block_diag.h, mat_block_dense.h);
 * Dense stores block (i,j) at block[i*numCols + j] (fac_helm2.c:914). */
typedef BfAbiMatBlock BfAbiMatBlockDiag;
typedef BfAbiMatBlock BfAbiMatBlockDense;

typedef struct BfAbiMatDenseComplex {
  BfAbiMat super;
  size_t rowStride;
  size_t colStride;
  double *data;              /* interleaved re,im (double _Complex) */
  void *pyArray;
} BfAbiMatDenseComplex;

typedef struct BfAbiMatDense {
  BfAbiMat super;
  void *vtable;              /* BfMatDenseVtable* */
  size_t rowStride;
  size_t colStride;
} BfAbiMatDense;

typedef struct BfAbiMatDenseReal {
  BfAbiMatDense super;
  double *data;
} BfAbiMatDenseReal;

typedef struct BfAbiMatIdentity {
  BfAbiMat super;
} BfAbiMatIdentity;

/* mat_sum.h:15-18: terms are summed (bfMatSumMul, src/mat_sum.c:54-83) */
typedef struct BfAbiMatSum {
  BfAbiMat super;
  BfAbiPtrArray termArr;
} BfAbiMatSum;

/* mat_coo_complex.h:19-30 */
typedef struct BfAbiMatCooComplex {
  BfAbiMat super;
  size_t numElts;
  size_t *rowInd;
  size_t *colInd;
  double *value;             /* interleaved re,im */
  size_t capacity;
} BfAbiMatCooComplex;

/* mat_diag_real.h:15-19 */
typedef struct BfAbiMatDiagReal {
  BfAbiMat super;
  size_t numElts;
  double *data;
} BfAbiMatDiagReal;

typedef struct BfAbiVec {
  BfAbiVecVtable *vtbl;
  int props;
  size_t size;
} BfAbiVec;

typedef struct BfAbiVecReal {
  BfAbiVec super;
  size_t stride;
  double *data;
} BfAbiVecReal;

typedef struct BfAbiVecComplex {
  BfAbiVec super;
  size_t stride;
  double *data;              /* interleaved re,im */
} BfAbiVecComplex;

/* typed views of the slots the engine calls */
typedef int    (*BfAbiGetTypeFn)(BfAbiMat const *);
typedef size_t (*BfAbiGetSizeFn)(BfAbiMat const *);
typedef BfAbiMat *(*BfAbiLikeFn)(BfAbiMat const *, size_t, size_t);
typedef void   (*BfAbiDeleteFn)(BfAbiMat **);
typedef void   (*BfAbiTransposeFn)(BfAbiMat *);            /* slot 63: in place, reference src/mat.c:271-273 */
typedef BfAbiMat *(*BfAbiMulFn)(BfAbiMat const *, BfAbiMat const *);
typedef BfAbiVec *(*BfAbiMulVecFn)(BfAbiMat const *, BfAbiVec const *);
typedef int    (*BfAbiVecGetTypeFn)(BfAbiVec const *);
typedef BfAbiVec *(*BfAbiVecCopyFn)(BfAbiVec const *);
typedef void   (*BfAbiVecDeleteFn)(BfAbiVec **);

/* ---- layout pins (SURVEY.md section 8(b) "ABI layout table") ------------- */
#if defined(__cplusplus)
#define BFABI_SA(c, m) static_assert(c, m)
#else
#define BFABI_SA(c, m) _Static_assert(c, m)
#endif
BFABI_SA(sizeof(BfAbiMat) == 32, "BfMat is 32 bytes");
BFABI_SA(offsetof(BfAbiMat, props) == 8, "BfMat.props@8");
BFABI_SA(offsetof(BfAbiMat, numRows) == 16, "BfMat.numRows@16");
BFABI_SA(offsetof(BfAbiMat, numCols) == 24, "BfMat.numCols@24");
BFABI_SA(sizeof(BfAbiMatVtable) == 528, "66 slots");
BFABI_SA(sizeof(BfAbiPtrArray) == 32, "BfPtrArray is 32 bytes");
BFABI_SA(offsetof(BfAbiPtrArray, num_elts) == 16, "BfPtrArray.num_elts@16");
BFABI_SA(sizeof(BfAbiMatProduct) == 64, "BfMatProduct is 64 bytes");
BFABI_SA(offsetof(BfAbiMatProduct, factorArr) == 32, "factorArr@32");
BFABI_SA(sizeof(BfAbiMatBlock) == 64, "BfMatBlock is 64 bytes");
BFABI_SA(offsetof(BfAbiMatBlock, block) == 40, "block@40");
BFABI_SA(offsetof(BfAbiMatBlock, rowOffset) == 48, "rowOffset@48");
BFABI_SA(offsetof(BfAbiMatBlock, colOffset) == 56, "colOffset@56");
BFABI_SA(sizeof(BfAbiMatBlockCoo) == 88, "BfMatBlockCoo is 88 bytes");
BFABI_SA(offsetof(BfAbiMatBlockCoo, numBlocks) == 64, "numBlocks@64");
BFABI_SA(offsetof(BfAbiMatBlockCoo, rowInd) == 72, "rowInd@72");
BFABI_SA(offsetof(BfAbiMatBlockCoo, colInd) == 80, "colInd@80");
BFABI_SA(sizeof(BfAbiMatDenseComplex) == 64, "BfMatDenseComplex is 64 bytes");
BFABI_SA(offsetof(BfAbiMatDenseComplex, rowStride) == 32, "rowStride@32");
BFABI_SA(offsetof(BfAbiMatDenseComplex, data) == 48, "data@48");
BFABI_SA(sizeof(BfAbiMatDense) == 56, "BfMatDense is 56 bytes");
BFABI_SA(offsetof(BfAbiMatDense, rowStride) == 40, "dense rowStride@40");
BFABI_SA(sizeof(BfAbiMatDenseReal) == 64, "BfMatDenseReal is 64 bytes");
BFABI_SA(offsetof(BfAbiMatDenseReal, data) == 56, "real data@56");
BFABI_SA(sizeof(BfAbiMatIdentity) == 32, "BfMatIdentity is 32 bytes");
BFABI_SA(sizeof(BfAbiMatSum) == 64, "BfMatSum is 64 bytes");
BFABI_SA(sizeof(BfAbiMatCooComplex) == 72, "BfMatCooComplex is 72 bytes");
BFABI_SA(offsetof(BfAbiMatCooComplex, value) == 56, "coo value@56");
BFABI_SA(sizeof(BfAbiMatDiagReal) == 48, "BfMatDiagReal is 48 bytes");
BFABI_SA(sizeof(BfAbiVec) == 24, "BfVec is 24 bytes");
BFABI_SA(sizeof(BfAbiVecReal) == 40, "BfVecReal is 40 bytes");
BFABI_SA(offsetof(BfAbiVecReal, stride) == 24, "vec stride@24");
BFABI_SA(offsetof(BfAbiVecReal, data) == 32, "vec data@32");

#ifdef __cplusplus
}
#endif
#endif /* BFHIP_ABI_H */
