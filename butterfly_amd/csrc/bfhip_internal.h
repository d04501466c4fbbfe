/* bfhip_internal.h -- shared between the C host (ir / plan / api) and the HIP
 * device layer (bfhip_device.hip).  The host side is plain C11; it reaches HIP
 * only through the `bfdev*` functions declared at the bottom (the "thin
 * C-ABI" between host C and device code). */
#ifndef BFHIP_INTERNAL_H
#define BFHIP_INTERNAL_H

#include <stddef.h>
#include <stdint.h>
#include "../../include/bfhip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------------
 * error plumbing
 * ---------------------------------------------------------------------- */
int bfhipFail(int code, char const *fmt, ...);   /* records message, returns code */

/* ------------------------------------------------------------------------
 * IR: owned copy of a BfhipDesc (or of a walked BfMat graph)
 * ---------------------------------------------------------------------- */
#define BF_REDUCE_SKIP 0xffffffffu
#define BF_LEAF_REAL 1u
#define BF_LEAF_CONJ 2u
typedef struct BfIr {
  uint32_t dtype;
  uint64_t numNodes, numChildren, root;
  uint8_t *kind;
  uint64_t *rows, *cols;
  uint64_t *childBegin;        /* numNodes+1 */
  uint64_t *childNode, *childRow0, *childCol0;
  void const **leafData;       /* borrowed host pointers (valid during compile only) */
  uint64_t *leafRowStride;
  uint64_t *leafColStride;     /* element stride between columns (BfMat graphs may have colStride != 1) */
  uint8_t *leafReal;           /* bit 0 (BF_LEAF_REAL): host values of this leaf are real doubles even in a complex operand (BfMatDiagReal
                                * terms); bit 1 (BF_LEAF_CONJ): the leaf is the CONJUGATE of the host values it points at (a dense complex
                                * leaf the reference flagged TRANS | CONJ: its strides are swapped as well) */
  uint64_t *synthBase;         /* per node: base index in the synthetic stream */
  int transposedView;          /* this IR is the transpose of the one the operator was compiled from (bfIrTransposed): synthetic leaf
                                * values are stream(base + col * rows + row), i.e. the ORIGINAL leaf's row-major index */
  uint64_t *topRowBlock;       /* per child of root, or NULL */
  uint32_t *depth;             /* stages needed by the subtree */
  /* sparse decorations folded into host-valued dense leaves: value added to leaf element (row, col) when the
   * arena is packed (sorted by leaf, row, col once finalized) */
  struct BfIrPatch *patches;
  uint64_t numPatches, capPatches;
  /* growable capacity (walker) */
  uint64_t capNodes, capChildren;
} BfIr;

typedef struct BfIrPatch { uint64_t leaf; uint32_t row, col; double re, im; } BfIrPatch;
void bfIrFree(BfIr *ir);
int bfIrFromDesc(BfhipDesc const *desc, BfIr *ir);
int bfIrFromBfMat(void const *bfMat, BfIr *ir);
int bfIrFinalize(BfIr *ir);   /* validation, depth, synthetic bases */
int bfIrTransposed(BfIr const *src, BfIr *dst);   /* dst = the expression of src^T over the same (borrowed) leaf values; src finalized */

/* ------------------------------------------------------------------------
 * Plan: the flattened per-stage layout (host mirror of what lives in HBM)
 * ---------------------------------------------------------------------- */

/* Where a vector segment lives.  All intermediates and partial-result slots
 * are sub-ranges of one device "vector arena" (element offsets, per RHS);
 * X and Y are the caller's buffers. */
enum { BF_SPACE_TEMP = 0, BF_SPACE_X = 1, BF_SPACE_Y = 2 };

/* device records (layouts shared with the kernels) */
typedef struct BfDevItem {
  uint32_t pieceBegin;
  uint32_t numPieces;
  uint32_t outOff;     /* element offset of row 0 of this item in its output space */
  uint32_t mrFlags;    /* bits 0..15: rows; bit 16: output space is Y (else vector arena) */
} BfDevItem;

typedef struct BfDevPiece {
  uint64_t dataOff;    /* element offset into the leaf arena (column-major mr_pad x ncols) */
  uint32_t inOff;      /* element offset of column 0 in the input space */
  uint32_t ncols;
  uint32_t flags;      /* bit 0: input space is X (else vector arena); bit 1: identity piece */
  uint32_t ld;         /* transposed plans: element stride between the lanes' columns (forward mrPad); row-major pieces: row stride; else 0 */
} BfDevPiece;

#define BF_ITEM_OUT_Y (1u << 16)
#define BF_COOP_BYTES (32u << 10)     /* transposed items at least this large are shared by the 4 wavefronts of a workgroup if they average BF_COOP_PIECES pieces */
#define BF_COOP_PIECES 8u
uint64_t bfPlanCountCoop(BfDevItem const *items, BfDevPiece const *pieces, uint64_t numItems, uint32_t elemSize);
#define BF_ITEM_TNARROW (1u << 20)    /* transposed plans: an item of <= 16 columns of a tall leaf (16-row-lane kernel); such items are the START of the list */
#define BF_ITEM_SMALL (1u << 19)      /* real, forward: <= 2 lane granules of rows, <= 16 pieces, <= BF_SMALL_COLS dense columns (one block);
                                        * small items are the END of a stage's item list and run four to a wavefront */
#define BF_SMALL_COLS 384u            /* (128 until round 4: the 8 KB row-major items of 130 - 380 columns of a streamed butterfly's stage 5 ran one per
                                        * wavefront at 5.56 TB/s; four to a wavefront 6.38) */
#define BF_SMALL_PIECES 16u
#define BF_ITEM_MERGED (1u << 18)     /* real, column-major: <= 64 pieces whose dense parts are ONE contiguous mrPad x n block, n <= BF_MERGE_COLS */
#define BF_MERGE_COLS 256u
#define BF_ITEM_ROWMAJOR (1u << 17)   /* all dense pieces of the item are stored row-major (few-row leaves of real operands) */
#define BF_PIECE_IN_X 1u
#define BF_PIECE_IDENTITY 2u
#define BF_PIECE_ROWMAJOR 4u          /* element (r, c) at dataOff + r * ld + c, ld = columns padded to the lane granule */

/* host-only: where each piece's values come from (for packing / synthesis) */
typedef struct BfPieceSrc {
  uint64_t node;       /* IR leaf */
  uint32_t row0, col0; /* sub-block origin inside the leaf */
} BfPieceSrc;

/* deterministic reduction of overlapping row groups (final accumulate into
 * Y, or any buffer several differently-shaped contributions land in) */
typedef struct BfReduce {
  uint32_t destSpace;       /* BF_SPACE_Y or BF_SPACE_TEMP */
  uint64_t destOff;         /* element offset in dest space */
  uint64_t numRows;
  uint64_t numIntervals;
  uint32_t *rowInterval;    /* [numRows] interval id of each row; BF_REDUCE_SKIP: the row is written directly by the one group that owns it */
  uint32_t *ivBegin;        /* [numIntervals+1] CSR into srcBias */
  int64_t *srcBias;         /* per source: (slot offset in arena) - (first row of the group) */
  uint64_t numSrc;
  uint32_t maxSrc;          /* longest source list of an interval (not stored in files: recomputed on load) */
  /* device copies */
  void *dRowInterval, *dIvBegin, *dSrcBias;
} BfReduce;

typedef struct BfStage {
  uint64_t numItems, numPieces;
  BfDevItem *items;
  BfDevPiece *pieces;
  BfPieceSrc *pieceSrc;
  uint32_t maxRows;          /* largest item row count */
  uint64_t firstSmall;       /* items [firstSmall, numItems) carry BF_ITEM_SMALL */
  uint64_t numNarrow;        /* transposed plans: items [0, numNarrow) carry BF_ITEM_TNARROW (<= 16 columns of A: their own launch) */
  uint32_t maxRowsRest, padRest;   /* largest item of the rest */
  uint64_t numCoopNarrow, numCoop;   /* the first items of either range that get a whole workgroup each (a scheduling hint) */
  uint64_t leafElems;        /* algorithmic: sum m*n over this stage's leaves */
  uint64_t vecIn, vecOut;    /* algorithmic vector elements read / written */
  uint64_t numReduce;
  BfReduce *reduce;
  /* dependency-driven launch (forward plans): which vector each piece reads / each item writes -- buffer ids of the
   * planner: 0 = x (pieces) or "nothing a later item waits for" (items: y, private slots), >= 2 an intermediate */
  uint32_t *pieceBuf;        /* [numPieces] */
  uint32_t *itemBuf;         /* [numItems] */
  /* forward complex128 plans: bundles of list neighbours that read the same input rows (bfPlanBundles) -- one workgroup of the
   * 64-RHS matrix-core kernel each.  bundleBegin[numBundles + 1]; the host copy is dropped with the other mirrors */
  uint32_t *bundleBegin;
  uint64_t numBundles;
  /* device copies */
  void *dItems, *dPieces, *dBundleBegin;
  /* persistent launches of a stage with more items than wavefront slots (bfStageKernelC128P): BF_TICKET_POOLS ticket
   * counters, zero between launches (the wavefront that draws a pool's last ticket of a launch puts it back to zero) */
  void *dTickets;
} BfStage;

typedef struct BfPlan {
  uint32_t dtype;            /* storage/compute type: BFHIP_C128 / F64 / F32 */
  uint32_t elemSize;         /* bytes per element */
  uint32_t epl;              /* elements per 16-byte lane load */
  uint32_t maxItemRows;      /* 64 * epl */
  uint32_t xcap;             /* max columns per piece (LDS staging capacity) */
  uint64_t numRows, numCols;
  uint64_t numStages;
  BfStage *stages;
  uint64_t arenaElems;       /* leaf arena size in elements */
  uint64_t tempElems;        /* vector arena elements per RHS */
  uint64_t numLeaves, leafElems;
  int transposed;            /* plan of A^T over the forward plan's arena */
  /* dependency-driven launch: number of vectors (buffer ids < numBufs), how many items write each, and whether the plan
   * qualifies (forward; every reduce pass sums into y, i.e. runs after all items) */
  uint64_t numBufs;
  uint32_t *bufWriters;      /* [numBufs] */
  int flowOk;
} BfPlan;

/* where the forward plan put each (leaf, row chunk, column range): the
 * transposed plan reads the same packed data with the roles of rows and
 * columns exchanged */
typedef struct BfFwdPiece {
  uint64_t node;
  uint64_t dataOff;
  uint32_t row0, mr, mrPad, col0, ncols;
  uint32_t rowMajor, ldr;      /* row-major forward piece and its row stride */
} BfFwdPiece;

typedef struct BfPlanOptions {
  uint32_t storeDtype;
  uint32_t itemRows;         /* rows per item cap (<= 64*epl); 0 -> default */
  uint32_t xcap;
  uint32_t groupByInput;     /* forward plans: order a stage's items by cost bucket, then by the input rows they read (RHS-block kernel: neighbours share an L2) */
  uint32_t itemsWanted;      /* a stage's groups are cut so that it has about this many items at least (0 -> 4096); never above 1 MiB per item */
  uint32_t minChunkRows;     /* lower bound of the adaptive item height, in 16-byte row units (0 -> 16): 32 keeps the RHS-block kernel's two-slab passes full */
  uint64_t rowBlockBegin, rowBlockEnd;
  uint64_t rowBegin, rowEnd;   /* row-range shard: keep what output rows [rowBegin, rowEnd) depend on; rowEnd == 0 -> all */
  /* transposed plan (A^T x): pieces are located in the forward plan's arena */
  BfFwdPiece const *fwdPieces;   /* sorted by (node, col0, row0); NULL -> forward plan */
  uint64_t numFwdPieces;
  uint32_t tCols;            /* transposed plan: columns of A per item, 16 (default) or 64 */
  uint32_t rowAlignBytes;    /* forward plan: start the rows of row-major pieces on this boundary (bfhipCompile: 128; 0 -> lane granule) */
} BfPlanOptions;

int bfPlanBuild(BfIr const *ir, BfPlanOptions const *po, BfPlan *plan);
#ifndef BF_BUNDLE_ITEMS
#define BF_BUNDLE_ITEMS 4u      /* wavefronts of the 64-RHS kernel's workgroup (A/B builds: 1 = no bundles) */
#endif
#define BF_BUNDLE_MIXED 0x80000000u   /* bundleBegin[] bit: four unrelated items (one-wavefront passes), not one shared X panel */
int bfPlanBundles(BfDevItem const *items, BfDevPiece const *pieces, uint64_t numItems, uint32_t **out, uint64_t *numBundles);
/* balanced contiguous row ranges for `world` ranks: cuts[world + 1], loads[world] (leaf elements each range keeps) or NULL */
int bfPlanRowPartition(BfIr const *ir, uint32_t world, uint64_t *cuts, uint64_t *loads);
/* table of the forward plan's pieces (needs its host mirrors); caller frees */
int bfPlanFwdPieces(BfPlan const *plan, BfFwdPiece **out, uint64_t *count);
void bfPlanFree(BfPlan *plan);

/* compile step with a caller-supplied arena fill (bfhip_build.c): `fill` runs
 * with the operator's device current, the plan's host mirrors still present
 * and the arena allocated; it must write every non-identity piece.  Consumes
 * `ir` like the plain compile. */
typedef int (*BfFillFn)(BfPlan const *plan, BfIr const *ir, void *dArena, void *ctx);
struct BfhipOptions;
struct BfhipOperator;
int bfhipCompileIrFill(BfIr *ir, struct BfhipOptions const *opts, BfFillFn fill, void *fillCtx, struct BfhipOperator **out);
/* HIP ordinal the operator lives on; -1 for a plan-only operator */
int bfhipOperatorDevice(struct BfhipOperator const *op);
/* vector arena for `nrhs` right-hand sides allocated now, so that applies of up to that many cannot fail on it */
int bfhipOperatorReserveRhs(struct BfhipOperator *op, uint32_t nrhs);

/* has the operator a plan of A^T (BFHIP_FLAG_ADJOINT / _PACKED); element type of the operand as given (BFHIP_C128 / BFHIP_F64) */
int bfhipOperatorHasAdjoint(struct BfhipOperator const *op);
uint32_t bfhipOperatorSrcDtype(struct BfhipOperator const *op);

/* GMRES around any device matvec (bfhip_gmres.c): `apply(ctx, dX, nrhs, dY, stream)` enqueues Y = A X on `stream`; n = order of A;
 * `device` = the HIP ordinal everything lives on.  bfhipSolveGMRESOptsDevice and bfhipShardedSolveGMRESDevice are this. */
typedef int (*BfGmresApplyFn)(void *ctx, void const *dX, size_t nrhs, void *dY, void *stream);
struct BfhipGmresOptions;
int bfGmresSolve(BfGmresApplyFn apply, void *ctx, uint64_t n, int device, struct BfhipGmresOptions const *opt, void const *dB, size_t nrhs,
                 void const *dX0, size_t *numIter, double *residual, void *dX, void *stream);

/* the sharded step on host vectors (bfhip_shard.hip): staging buffers of the sharded object; used by the vtable shim */
struct BfhipSharded;
int bfhipShardedApplyHost(struct BfhipSharded *sh, int transpose, void const *X, size_t ldx, size_t nrhs, void *Y, size_t ldy);
struct BfhipOperator *bfhipShardedOperator(struct BfhipSharded const *sh);

/* ------------------------------------------------------------------------
 * Device layer (implemented in bfhip_device.hip)
 * ---------------------------------------------------------------------- */
int bfdevSetDevice(int device);                 /* -1: keep current; returns BfError */
int bfdevGetDevice(int *device);
int bfdevMalloc(void **p, size_t bytes);
void bfdevFree(void *p);
int bfdevMemcpyH2D(void *dst, void const *src, size_t bytes);
int bfdevMemcpyD2H(void *dst, void const *src, size_t bytes);
int bfdevMemset(void *dst, int value, size_t bytes);
int bfdevSync(void *stream);
int bfdevPointerKind(void const *p);            /* 0 pageable host, 1 this device, 2 pinned / registered host, 3 another device */
int bfdevHostRegister(void *p, size_t bytes);
int bfdevHostUnregister(void *p);
int bfdevMemcpyAnyAsync(void *dst, void const *src, size_t bytes, void *stream);
int bfdevHostAllocPinned(void **p, size_t bytes);
void bfdevHostFreePinned(void *p);

/* fill pieces [p0, p1) of a stage with the synthetic stream, directly in HBM */
typedef struct BfSynthPiece {
  uint64_t dataOff;     /* element offset in arena */
  uint64_t vbase;       /* synthetic index of leaf element (0,0) */
  uint32_t strideR, strideC;   /* virtual index of leaf element (i, j) = vbase + i * strideR + j * strideC (n, 1 -- or 1, rows of the leaf
                                * as stored, for the leaves of a transposed view) */
  uint32_t row0, col0;
  uint32_t mr, mrPad, ncols;
  uint32_t rowMajor, ldr;      /* row-major piece: element (r, c) at dataOff + r * ldr + c */
  double scale;
} BfSynthPiece;
int bfdevSynthFill(void *arena, uint32_t dtype, BfSynthPiece const *hostPieces, uint64_t count, uint64_t seed);

typedef struct BfLaunchArgs {
  void const *arena;
  void const *items;
  void const *pieces;
  uint64_t numItems;
  uint64_t firstSmall;   /* == numItems when the stage has no small items */
  uint64_t numNarrow, numCoopNarrow, numCoop;      /* transposed: see BfStage */
  uint32_t maxRowsRest;
  void const *x;
  void *y;
  void *temp;
  void const *zero;      /* device buffer of >= 64 zero bytes */
  uint32_t nrhs;
  uint32_t dtype;
  uint32_t maxRows;
  int transposed;        /* pieces carry `ld`: lanes own columns of the forward pieces */
  void *tickets;         /* NULL, or BF_TICKET_POOLS x BF_TICKET_STRIDE uint32 owned by this stage, zero between launches (see BfStage.dTickets) */
  uint32_t exactComplex; /* BFHIP_FLAG_EXACT_COMPLEX: the matrix-core kernels form complex products with four real multiplications */
  uint32_t pad2;
  void const *bundles;   /* forward complex128: BfStage.dBundleBegin (NULL: none) */
  uint64_t numBundles;
} BfLaunchArgs;
#define BF_TICKET_POOLS 64u
#define BF_TICKET_STRIDE 64u      /* uint32 between two pools' counters: a 256-byte block each -- counters that share a cache line share its atomic unit (measured: 64 packed counters behaved like one) */
int bfdevLaunchStage(BfLaunchArgs const *a, void *stream);
/* bfhip_persist.hip (experimental persistent launch of the complex128 stage kernel) */
uint32_t bfdevPersistentGrid(void);
int bfdevLaunchPersistC128(void const *stageParams, uint32_t grid, void *tickets, void *timeline, void *stream);

/* dependency-driven launch of a whole forward complex128 plan (bfFlowKernelC128): see bfhip_device.hip */
typedef struct BfFlowArgs {
  void const *arena;
  void const *items, *pieces, *itemOut, *writers;   /* flat over all stages; pieces' `ld` holds the vector id they read */
  void *counters;        /* uint32[numBufs]: [0] the ticket queue, [1] the error flag, [id >= 2] writes seen by vector id */
  uint32_t numItems, nrhs, epoch, queueBase, gridWorkgroups;
  void const *x;
  void *y, *temp;
} BfFlowArgs;
int bfdevLaunchFlow(BfFlowArgs const *a, void *stream);
/* EXPERIMENTAL builds: timeline / persistent launch of a complex128 stage (bfhip_experimental.hip); *handled = 0 -> the caller launches it */
int bfdevLaunchStageExperimental(BfLaunchArgs const *a, void const *stageParams, uint32_t grid, void *stream, int *handled);
int bfdevFlowGrid(uint64_t numItems, uint32_t *grid);
int bfdevMemsetAsync(void *dst, int value, size_t bytes, void *stream);

typedef struct BfReduceArgs {
  void const *rowInterval, *ivBegin, *srcBias;
  uint64_t numRows;
  void const *temp;
  void *dest;            /* already offset to destOff*nrhs */
  uint32_t nrhs;
  uint32_t dtype;
  uint32_t longLists;    /* some row has >= 64 partial sums: the launch uses the kernel that loads them 32 at a time */
  uint32_t pad;
} BfReduceArgs;
/* all `count` reduces (one stage; same temp / nrhs / dtype) in as few launches as possible */
int bfdevLaunchReduce(BfReduceArgs const *a, uint32_t count, void *stream);
int bfdevScalePermute(void *dst, void const *src, void const *scale, int power, uint64_t const *perm, uint64_t n, uint32_t dtype, void *stream);

/* device-resident GMRES building blocks (complex128; bfhip_gmres.c drives them).
 * Vectors are n x nrhs row-major; reductions are per RHS column, two-stage and
 * in fixed order (per-block partials, then a tree over the partials), so a
 * solve is bit-reproducible.  `nb` = number of row blocks = partials per RHS. */
int bfdevGmresResidual(void const *B, void const *AX0, void *W, void *partialOut, uint64_t n, uint32_t nrhs, uint32_t nb, void *stream);
int bfdevGmresDot(void const *Vi, void const *W, void *partialOut, uint64_t n, uint32_t nrhs, uint32_t nb, void *stream);
/* h = sum(partialIn); hOut[q] = h; W -= h * Vi; then partialOut = conj(Vnext).W (Vnext != NULL) or |W|^2 */
int bfdevGmresMgsStep(void const *Vi, void const *Vnext, void *W, void const *partialIn, void *partialOut, void *hOut,
                      uint64_t n, uint32_t nrhs, uint32_t nb, void *stream);
/* batched Gram-Schmidt pass over V_0..V_{numVec-1} (vectors n*nrhs apart): partial[(q*numVec + i)*nb + b] */
int bfdevGmresDots(void const *V, void const *W, void *partial, uint64_t n, uint32_t nrhs, uint32_t nb, uint32_t numVec, void *stream);
/* h[i*nrhs + q] = sum_b partial; hSum = h + hPrev (hSum / hPrev may be NULL) */
int bfdevGmresDotsFinish(void const *partial, void const *hPrev, void *h, void *hSum, uint32_t nrhs, uint32_t nb, uint32_t numVec, void *stream);
/* W -= sum_i h_i V_i; partialOut (may be NULL) = per-block |W|^2 */
int bfdevGmresProject(void const *V, void *W, void const *h, void *partialOut, uint64_t n, uint32_t nrhs, uint32_t nb, uint32_t numVec, void *stream);
/* nrm = sqrt(sum(partialIn)); hOut[q] = nrm; Vout = W / nrm */
int bfdevGmresFinish(void const *W, void const *partialIn, void *Vout, void *hOut, uint64_t n, uint32_t nrhs, uint32_t nb, void *stream);
/* X = X0 + sum_{i<j} V_i * y[i] ; V = (j) vectors of n*nrhs, y = [j][nrhs] */
int bfdevGmresUpdate(void const *X0, void const *V, void const *y, uint32_t j, void *X, uint64_t n, uint32_t nrhs, void *stream);
int bfdevMemcpyD2HAsync(void *dst, void const *src, size_t bytes, void *stream);

/* ------------------------------------------------------------------------
 * Builder device layer (bfhip_build.hip; driven by bfhip_build.c).  All
 * matrices are complex128, column-major; offsets are in complex elements
 * relative to the base pointer named per call.
 * ---------------------------------------------------------------------- */
typedef struct BfBuildPts {        /* = BfhipPointSet (include/bfhip_build.h) */
  uint32_t kind, count;
  uint64_t first;
  double cx, cy, r;
} BfBuildPts;

/* one kernel matrix to evaluate: dst[i + j*tgt.count] = G(tgt_i, src_j) */
typedef struct BfEvalMat {
  BfBuildPts src, tgt;
  void *dst;
  uint32_t pot;          /* 0: S, 1: S' (target normals; tree targets), 2: D (source normals), 3: alpha S + beta D */
  uint32_t decorate;     /* 1: apply column weights / self value (leaves of the operator, not Z_equiv) */
} BfEvalMat;
/* what every kernel evaluation needs besides the two points */
typedef struct BfEvalEnv {
  void const *dPoints, *dNormals, *dColWeights;   /* device; normals / weights may be NULL */
  void const *dTgtPoints, *dTgtNormals;           /* separate target tree (BFHIP_PTS_TREE_TGT) or NULL */
  void const *dOrigIndex;                         /* device uint64[numPoints] or NULL */
  double wavenumber, selfRe, selfIm;
  double alphaRe, alphaIm, betaRe, betaIm;
  uint64_t numPoints;
  uint32_t krOrder;                               /* 0, 2, 6, 10 */
  unsigned long long *dKrHits;                    /* device counter of corrected entries (consistency check) or NULL */
} BfEvalEnv;
/* tilePrefix[numMats+1]: prefix sums of ceil(rows*cols / BF_EVAL_TILE) */
#define BF_EVAL_TILE 1024u
int bfdevBuildEval(BfEvalMat const *hostMats, uint64_t const *hostTilePrefix, uint64_t numMats, BfEvalEnv const *env);

/* one-sided Jacobi SVD of A (mt x me, mt >= me), in place: on return the
 * columns of A are U*Sigma, V (me x me) holds the right singular vectors and
 * scale[j] = 1/sigma_j^2 (0 for truncated sigma_j < max(mt,me) eps sigma_max + eps) */
typedef struct BfSvdProb {
  void *a, *v;
  double *scale;         /* [me] */
  uint32_t mt, me;
  uint32_t dim, pad;     /* the max(rows, cols) of the truncation rule (src/mat_dense_complex.c:1800-1812): that of the ORIGINAL
                            matrix when (a) is the QR-preconditioned one */
} BfSvdProb;
typedef struct BfSvdStats { unsigned long long maxSweeps, notConverged, truncated, sumSweeps; } BfSvdStats;
int bfdevBuildJacobi(BfSvdProb const *hostProbs, uint64_t numProbs, BfSvdStats *stats);

/* QR with column pivoting ahead of the Jacobi SVD (bfQrcpKernel): a (mt x me, ld mt) is overwritten (R in its upper
 * triangle), b (mt x n, ld mt) becomes Q^H b, x (me x me workspace) receives the me x rank matrix the Jacobi kernel
 * then works on (ld me); hostRanks[i] = the number of steps taken before the largest remaining column fell below
 * dim * eps * (largest column of a). */
typedef struct BfQrProb {
  void *a, *b, *x;
  uint32_t mt, me, n, dim;
} BfQrProb;
int bfdevQrcpFits(uint32_t mt, uint32_t me);
int bfdevBuildQrcp(BfQrProb const *hostProbs, uint64_t numProbs, uint32_t *hostRanks);

/* C (M x N) = op(A) * B with optional row scaling C[i,:] *= scale[i];
 * transA: op(A)[i,k] = conj(A[k + i*lda]) (A stored K x M), else A[i + k*lda] */
typedef struct BfGemmJob {
  void const *a, *b;
  void *c;
  double const *scale;   /* NULL: none */
  uint32_t M, N, K, lda, ldb, ldc, transA, pad;
} BfGemmJob;
int bfdevBuildGemm(BfGemmJob const *hostJobs, uint64_t numJobs);

/* arena[dataOff + c*mrPad + r] = r < mr ? store[srcOff + c*srcLd + r] : 0 */
typedef struct BfPackPiece {
  uint64_t dataOff, srcOff;
  uint32_t srcLd, mr, mrPad, ncols;
} BfPackPiece;
int bfdevBuildPack(void *arena, void const *store, BfPackPiece const *hostPieces, uint64_t count);

int bfdevMemFree(uint64_t *freeBytes);    /* free device memory right now */

/* y = G x, N x N single-layer kernel evaluated on the fly; scratch is allocated inside */
/* numTgt == 0: square (targets = the points); else targets = env->dTgtPoints */
int bfdevHelm2Dense(BfEvalEnv const *env, uint32_t pot, uint64_t n, uint64_t numTgt, void const *dX, void *dY, void *stream);
int bfdevMemcpyH2DAsync(void *dst, void const *src, size_t bytes, void *stream);
int bfdevMemcpyD2DAsync(void *dst, void const *src, size_t bytes, void *stream);

/* events for BFHIP_FLAG_PROFILE */
int bfdevEventCreate(void **ev);
void bfdevEventDestroy(void *ev);
int bfdevEventRecord(void *ev, void *stream);
int bfdevEventElapsed(void *start, void *stop, float *ms);
int bfdevEventSync(void *ev);

#ifdef __cplusplus
}
#endif
#endif
