/* bfhip.h -- C-ABI of the MI355X butterfly-apply engine (libbfhip.so).
 *
 * One hot path of sampotter/butterfly is replaced: applying an already-built
 * factorization, i.e. `bfMatMul(A, X)` / `bfMatMulVec(A, x)` where A is a
 * `BfMatProduct` / `BfMatBlock{Dense,Diag,Coo}` hierarchy over dense leaves
 * (reference src/mat.c:183-189 dispatching to src/mat_product.c:211-280,
 * src/mat_block_dense.c:512-638, src/mat_block_diag.c:370-456,
 * src/mat_block_coo.c:382-474, src/mat_dense_complex.c:1024-1051,
 * src/mat_dense_real.c:1373-1407, src/mat_identity.c:149-181).
 *
 * Usage mirrors what a cgo/ctypes/C caller of the reference would bind:
 *
 *   BfhipOperator *op;
 *   bfhipCompile(A, NULL, &op);          // walk the BfMat graph once, upload
 *   BfMat *A_hip = bfhipMatNew(op);      // drop-in BfMat: bfMatMul(A_hip, X)
 *   ... or bfhipApply(op, X, ldx, nrhs, Y, ldy) on raw host buffers,
 *   ... or bfhipApplyDevice(op, dX, nrhs, dY, stream) on resident vectors.
 *   bfhipFree(&op);
 *
 * Every function returns 0 (BF_ERROR_NONE) or a reference `enum BfError`
 * value (include/bf/error.h:3-16); nothing in this library aborts.  All
 * pointers are plain C; no torch / HIP types appear in signatures (streams
 * are passed as `void *` = hipStream_t).
 *
 * Thread-safety: as the reference (single host thread per operator).
 */
#ifndef BFHIP_H
#define BFHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct BfhipOperator BfhipOperator;

/* element type of leaves and vectors */
enum {
  BFHIP_C128 = 0,  /* double _Complex: fac_helm2 operands (BfMatDenseComplex) */
  BFHIP_F64 = 1,   /* double: fac_streamer operands (BfMatDenseReal)          */
  BFHIP_F32 = 2    /* float: build extension -- real leaves demoted on upload */
};

/* node kinds of the flat expression descriptor */
enum {
  BFHIP_NODE_DENSE = 0,     /* leaf: rows x cols row-major values          */
  BFHIP_NODE_IDENTITY = 1,  /* leaf: square identity (mat_identity.c:149)   */
  BFHIP_NODE_BLOCK = 2,     /* sum of children placed at (row0, col0)      */
  BFHIP_NODE_PRODUCT = 3    /* F0 * F1 * ... * F_{L-1}, applied right-to-left */
};

enum {
  BFHIP_FLAG_NONE = 0,
  BFHIP_FLAG_PROFILE = 1u << 0,  /* record hipEvents around every stage launch */
  BFHIP_FLAG_PLAN_ONLY = 1u << 1,/* build the host-side plan only: no device is touched, apply is refused;
                                    for inspecting the flattened layout (bfhipPlan* below) */
  BFHIP_FLAG_ADJOINT = 1u << 2,  /* also build the plan of A^T (bfhipApplyTranspose*, RmulVec of the shim):
                                    index metadata only, the packed leaf data is shared */
  BFHIP_FLAG_ADJOINT_PACKED = 1u << 4,  /* as BFHIP_FLAG_ADJOINT, but A^T gets a packed copy of the leaves of its own: the adjoint
                                    plan is then a FORWARD plan of the transposed expression (factors reversed, every leaf
                                    transposed -- what bfMatProductTranspose leaves behind, src/mat_product.c:409-420) and
                                    A^T x runs on the forward kernels at the forward rate, for twice the leaf memory.
                                    Falls back to the shared-leaf plan (BFHIP_FLAG_ADJOINT) with a device value builder (bfhipBuildHelm2: its
                                    values exist in the forward arena only), on a row shard (rowBegin / rowBlockBegin: the shard's adjoint is the
                                    pruned transposed task list) and when the second arena does not fit the device's memory; a plan-only operator
                                    keeps the packed plan (bfhipPlanPackArenaT packs its arena).  bfhipSave / bfhipLoad carry both arenas */
  BFHIP_FLAG_EXACT_COMPLEX = 1u << 5,  /* complex128 operators applied to blocks of right-hand sides (nrhs >= 2, the matrix-core kernels): form
                                    every complex product with its FOUR real multiplications, as cblas_zgemm's recurrence does
                                    (src/mat_dense_complex.c:1704-1765), instead of Gauss's three.  The default is normwise as accurate
                                    (same 1e-12 tolerance against the oracle) but its imaginary part carries a rounding error of
                                    eps * sum (|Ar| + |Ai|)(|Xr| + |Xi|) whatever its own size; with this flag each part's error is
                                    bounded by eps times ITS OWN sum of absolute products (e.g. nearly real data keep a relatively
                                    accurate small imaginary part).  Costs a third more matrix-pipe work (nrhs = 64: ~25 % slower).
                                    One right-hand side (the GEMV kernel) always uses the four-product form */
  BFHIP_FLAG_FLOW = 1u << 3      /* EXPERIMENTAL BUILDS ONLY (libbfhip_exp.so, `make -C butterfly_amd/csrc experimental`): complex128
                                    operators applied to ONE right-hand side run the whole plan as ONE dependency-driven
                                    persistent launch (items wait for the intermediate vectors they read, not for the previous
                                    stage; bit-identical results).  Measured slower than the staged launches on MI355X
                                    (DESIGN_EXPERIMENTS.md section 15).  The product library (libbfhip.so) does not contain that executor
                                    and refuses the flag with BF_ERROR_NOT_IMPLEMENTED */
};
/* Environment variables.  The product library reads six, all test / diagnosis hooks:
 *   BFHIP_JACOBI_GLOBAL=1   builder: every SVD problem through the global-memory fallback kernel (test hook)
 *   BFHIP_JACOBI_QR_MIN=n   builder: least-squares problems of >= n equivalent sources are QR-factored with column pivoting
 *                           before the Jacobi SVD (default 65; 0: every problem, a test hook; a huge value: none)
 *   BFHIP_JACOBI_GRAM_MIN=n builder: least-squares problems of rows + columns >= n run the block form of the Jacobi SVD on Gram matrices
 *                           (default 512; 0: every problem, a test hook; a huge value: none)
 *   BFHIP_JACOBI_PROFILE=1  builder: prints the time of every phase and of every SVD size class to stderr (synchronises per class)
 *   BFHIP_ALLOW_UNCONVERGED_SVD=1   builder: keep an operator whose Jacobi SVDs hit the sweep limit (diagnosis)
 *   BFHIP_GMRES_MGS=1       GMRES: the reference's modified Gram-Schmidt order instead of batched CGS2 (as the per-call option)
 * The experimental build (libbfhip_exp.so) additionally reads, once per process:
 *   BFHIP_FLOW=1            as BFHIP_FLAG_FLOW for every operator compiled in the process
 *   BFHIP_PERSISTENT=1      complex128 stages with more items than wavefront slots run as a persistent grid that draws
 *                           pooled tickets (bfhip_persist.hip; bit-identical results, measured equal to slower)
 *   BFHIP_TIMELINE_FILE=p   every complex128 stage launch (one right-hand side) becomes a synchronous diagnostic launch
 *                           that appends each item's start / end stamps to file p (tools/timeline.py; same results)
 *   BFHIP_FLOW_SPIN, BFHIP_FLOW_DEBUG, BFHIP_FLOW_DEBUGMODE   diagnostics of the one-launch executor */

typedef struct BfhipOptions {
  uint32_t structSize;      /* = sizeof(BfhipOptions) */
  int32_t device;           /* HIP ordinal; -1 = current device */
  uint32_t flags;           /* BFHIP_FLAG_* */
  uint32_t maxRhs;          /* intermediates preallocated for this many RHS (0 -> 1); grows on demand */
  uint32_t demoteToF32;     /* 1: store/compute BFHIP_F64 operands in fp32 (config 5 extension) */
  uint32_t reserved0;
  uint64_t seed;            /* value seed for synthetic (data == NULL) leaves */
  /* row sharding: keep only block rows
   * [rowBlockBegin, rowBlockEnd) of a BLOCK root; rowBlockEnd == 0 -> all.
   * The operator then maps the full x to the concatenation of those rows. */
  uint64_t rowBlockBegin;
  uint64_t rowBlockEnd;
  /* row-range sharding: keep exactly what output rows [rowBegin, rowEnd) depend on (any operand, not only a BLOCK
   * root): leaves no kept row needs are dropped, source-side factors several ranges need are replicated, and the
   * rows produced are bit for bit those of the whole operator when no leaf straddles the range ends (the cuts
   * bfhipRowPartition returns).  rowEnd == 0 -> all rows.  Exclusive with rowBlockBegin/End.  (Fields added in
   * round 3: a caller passing the shorter round-2 struct through structSize gets "all rows".) */
  uint64_t rowBegin;
  uint64_t rowEnd;
} BfhipOptions;
#define BFHIP_OPTIONS_SIZE_V1 48u   /* sizeof(BfhipOptions) before rowBegin / rowEnd */

/* Flat description of an operand: the same expression tree as a BfMat graph,
 * as arrays.  Used (a) internally by bfhipCompile after walking a BfMat graph
 * and (b) directly by callers that synthesize structure-exact operands
 * without ever materializing host values (leafData[i] == NULL: values are
 * generated on the device from `seed`, see bfhipSyntheticValue). */
typedef struct BfhipDesc {
  uint32_t structSize;      /* = sizeof(BfhipDesc) */
  uint32_t dtype;           /* BFHIP_C128 or BFHIP_F64 */
  uint64_t numNodes;
  uint64_t root;
  const uint8_t *kind;          /* [numNodes] BFHIP_NODE_* */
  const uint64_t *rows;         /* [numNodes] element rows */
  const uint64_t *cols;         /* [numNodes] element cols */
  const uint64_t *childBegin;   /* [numNodes+1] CSR into child arrays */
  const uint64_t *childNode;    /* [numChildren] */
  const uint64_t *childRow0;    /* [numChildren] row offset inside parent (BLOCK) */
  const uint64_t *childCol0;    /* [numChildren] col offset inside parent (BLOCK) */
  const void *const *leafData;  /* [numNodes] or NULL; row-major host values */
  const uint64_t *leafRowStride;/* [numNodes] or NULL (-> cols), in elements */
  const uint64_t *topRowBlock;  /* [numChildren of root] or NULL: block-row id of each root child (for sharding) */
  const uint8_t *blockKind;     /* [numNodes] or NULL: reference BfType of BLOCK nodes (15 coo, 16 dense, 17 diag); informative only */
} BfhipDesc;

typedef struct BfhipStats {
  uint32_t structSize;
  uint32_t dtype;
  uint64_t numRows, numCols;
  uint64_t numStages;
  uint64_t numLeaves;        /* dense leaves kept (after sharding) */
  uint64_t numItems;         /* kernel work items over all stages */
  uint64_t numPieces;
  uint64_t leafElems;        /* sum over leaves of m*n */
  uint64_t leafBytes;        /* leafElems * sizeof(element) = algorithmic operand bytes */
  uint64_t vecElemsRead;     /* sum over stages of input elements read once per distinct segment */
  uint64_t vecElemsWritten;  /* sum over stages of output elements */
  uint64_t arenaBytes;       /* device bytes actually held for leaves (incl. padding) */
  uint64_t tempElems;        /* intermediate-vector elements per RHS */
  uint64_t metaBytes;        /* device bytes of index metadata */
} BfhipStats;

/* ---- compile ------------------------------------------------------------- */

/* Walk a reference object graph (read-only; nothing of A is retained) and
 * build the device operator.  `bfMat` is a `BfMat const *` of the reference
 * (layouts: bfhip_abi.h).  Replaces the recursive dispatch the reference does
 * on every bfMatMul call (src/mat.c:183 and callees listed above).
 * Errors: TYPE_ERROR (unknown node type / mixed real+complex),
 * NOT_IMPLEMENTED (a dense leaf flagged CONJ without TRANS; a leaf flagged TRANS is read as the reference multiplies by
 * it: a complex one as its conjugate transpose, mat_dense_complex.c:27-35, 503-511, a real one as its transpose),
 * INVALID_ARGUMENTS (NULL vtable, non-monotone offsets, shape mismatch),
 * MEMORY_ERROR (host or device OOM), RUNTIME_ERROR (HIP failure). */
int bfhipCompile(const void *bfMat, const BfhipOptions *opts, BfhipOperator **out);

/* Same, from a flat descriptor. */
int bfhipCompileDesc(const BfhipDesc *desc, const BfhipOptions *opts, BfhipOperator **out);

/* Balanced contiguous row ranges for `world` ranks: cuts[0] = 0 < ... < cuts[world] = rows, rank r owning rows
 * [cuts[r], cuts[r + 1]) (BfhipOptions.rowBegin/rowEnd).  Cuts fall only where no leaf that writes y straddles them
 * (quadtree node boundaries of a fac_helm2 operand: the level-3 and deeper row blocks inside each level-2 BlockDense,
 * reference src/fac_helm2.c:814-858), and the largest per-rank load -- leaf elements kept, replicated source-side
 * factors included -- is minimised.  `leafElems` [world] may be NULL.  INVALID_ARGUMENTS if the operand offers fewer
 * than `world` ranges.  No device is touched. */
int bfhipRowPartition(const BfhipDesc *desc, uint32_t world, uint64_t *cuts, uint64_t *leafElems);
int bfhipRowPartitionMat(const void *bfMat, uint32_t world, uint64_t *cuts, uint64_t *leafElems);

/* leafElems[v] = sum of rows x cols over the dense leaves under node v, for every node of the descriptor (what
 * bfMatNumBytes / 16 resp. / 8 of that subtree's leaves is): balancing ranks, sampling sub-operators. */
int bfhipDescSubtreeLeafElems(const BfhipDesc *desc, uint64_t *leafElems);

/* ---- apply --------------------------------------------------------------- */

/* Y[numRows x nrhs] = A * X[numCols x nrhs]; host buffers, row-major with
 * leading dimensions ldx/ldy in elements (the layout of BfMatDenseComplex
 * with colStride == 1, mat_dense_complex.h:69-81).  Synchronous. */
int bfhipApply(BfhipOperator *op, const void *X, size_t ldx, size_t nrhs, void *Y, size_t ldy);

/* What bfhipApply (and every slot of the vtable shim) does with X and Y depends on what they are (hipPointerGetAttributes):
 * densely packed (ld == nrhs) double-precision vectors in DEVICE memory of the operator's GPU are used in place, in PINNED or
 * REGISTERED host memory they are the source / target of the DMA itself, anything else is packed through the operator's pinned
 * staging buffer (one more pass over the vector on the CPU, each way).  The library never registers a caller's buffer on its own
 * (it cannot know when the caller frees it); a caller that reuses its vectors -- the Krylov basis of bfSolveGMRES, say --
 * registers them once (hipHostRegister underneath; undo before freeing the memory). */
int bfhipHostRegister(void *p, size_t bytes);
int bfhipHostUnregister(void *p);

/* Same on device-resident, densely packed (ld == nrhs) vectors; enqueued on
 * `stream` (hipStream_t, NULL = default stream) and asynchronous. */
int bfhipApplyDevice(BfhipOperator *op, const void *dX, size_t nrhs, void *dY, void *stream);

/* Y[numCols x nrhs] = A^T X[numRows x nrhs] (plain transpose, no conjugation): what the reference's
 * bfMatRmulVec computes on a BfVecReal (x^T A as a vector: src/mat_product.c:314-345,
 * src/mat_block_diag.c:458-505, src/mat_block_coo.c:476-520, src/mat_block_dense.c:696-758,
 * src/mat_dense_real.c:1508-1542) and what `cov_matvec` needs for Phi^T v
 * (examples/covariance/lbo_cov.c:48-60).  Needs BFHIP_FLAG_ADJOINT at compile. */
int bfhipApplyTranspose(BfhipOperator *op, const void *X, size_t ldx, size_t nrhs, void *Y, size_t ldy);
int bfhipApplyTransposeDevice(BfhipOperator *op, const void *dX, size_t nrhs, void *dY, void *stream);

/* ---- GMRES (the production caller of the apply path) --------------------- */

/* Solve A X = B with the operator as A, mirroring the reference's
 * bfSolveGMRES(A, B, X0, tol, maxNumIter, &numIter, M = NULL)
 * (src/linalg.c:47-317): unrestarted GMRES, modified Gram-Schmidt, Givens
 * rotations, convergence when max_p |s_{j+1,p}| / max_p ||r_p|| < tol.  The
 * Krylov basis, the work vector, x0 and b stay on the device; only the new
 * Hessenberg column crosses PCIe per iteration.  X0 may be NULL (zeros).
 * `numIter` receives the reference's iteration count (the number of basis
 * vectors the solution is built from), `residual` the last relative residual;
 * either may be NULL.  Complex square operators only; the left preconditioner
 * (the reference's M) is the *Precond* entry below.  Host-pointer form: B, X0, X are
 * row-major n x nrhs with leading dimensions in elements. */
int bfhipSolveGMRES(BfhipOperator *op, const void *B, size_t ldb, size_t nrhs, const void *X0, size_t ldx0,
                    double tol, size_t maxNumIter, size_t *numIter, double *residual, void *X, size_t ldx);
/* Device-resident form (densely packed n x nrhs buffers on the operator's
 * GPU, which must be the current device); synchronous on return. */
int bfhipSolveGMRESDevice(BfhipOperator *op, const void *dB, size_t nrhs, const void *dX0, double tol,
                          size_t maxNumIter, size_t *numIter, double *residual, void *dX, void *stream);

/* ---- multi-GPU: one process per GPU, one RCCL collective per apply -------- */
/* The top-level block rows of a factorization are independent (bfMatBlockDenseMul computes each from the
 * full x, src/mat_block_dense.c:534-566), so they shard across the GPUs of a node with no data-path
 * exchange except the closing collective on y.  Each rank compiles the operator of
 * ITS share (BfhipOptions.rowBlockBegin/End, or a descriptor restricted to its blocks) and then:
 *
 *   char id[128]; if (rank == 0) bfhipCommGetUniqueId(id);  ... ship `id` to every rank (MPI, a file, ...)
 *   bfhipCommInitRank(id, nranks, rank, device, &comm);      // ncclCommInitRank
 *   bfhipShardedCreate(op, comm, &spec, maxRhs, &sh);
 *   bfhipShardedApplyDevice(sh, dX, nrhs, dY, stream);       // every rank: x replicated in, FULL y out
 *
 * BFHIP_SHARD_ROWS:   this rank's operator yields the rows of the segments it owns, in global order,
 *                     compacted; the step ends with ONE in-place ncclAllGather (xGMI) and a small kernel
 *                     that puts the segments into global row order in dY.
 * BFHIP_SHARD_BLOCKS: this rank's operator yields a full-length partial y (its (row, col) blocks at
 *                     their original offsets, zeros elsewhere); the step ends with ONE ncclAllReduce.  The
 *                     summation order is RCCL's: equal to one GPU to rounding, not bit for bit (ROWS is).
 * Everything is enqueued on `stream` behind the stage kernels; nothing synchronizes the host.  RCCL is
 * looked up at run time (the copy already in the process, else librccl.so.1); without it these entry
 * points return RUNTIME_ERROR and the rest of the library works. */
typedef struct BfhipComm BfhipComm;
typedef struct BfhipSharded BfhipSharded;
enum { BFHIP_SHARD_ROWS = 0, BFHIP_SHARD_BLOCKS = 1 };
typedef struct BfhipShardSpec {
  uint32_t structSize;      /* = sizeof(BfhipShardSpec) */
  uint32_t mode;            /* BFHIP_SHARD_* */
  uint64_t numRowsGlobal;   /* rows of the whole operator = length of y */
  uint32_t numSegments;     /* ROWS: row segments of y (top-level block rows, row ranges) */
  uint32_t reserved;
  const uint64_t *segRows;  /* [numSegments] rows of each segment */
  const uint32_t *segOwner; /* [numSegments] rank that computes it */
  /* ROWS, optional (round 3; NULL or a struct of the shorter round-2 size: segments are consecutive in global order,
   * each row computed by exactly one rank).  [numSegments] first global row of each segment.  Several segments may then
   * cover the SAME rows (identical ranges only): ranks that share a block row by columns each send a partial result
   * for it, and after the one all-gather every rank adds the partials of a range in LIST order -- a fixed order, so the
   * result is the same on every rank and from run to run (to rounding the one-GPU result, not bit for bit: that is the
   * price of splitting a row's sum).  A rank's operator yields its segments in list order, compacted. */
  const uint64_t *segGlobalOff;
} BfhipShardSpec;
#define BFHIP_SHARDSPEC_SIZE_V1 40u   /* sizeof(BfhipShardSpec) before segGlobalOff */
int bfhipCommGetUniqueId(void *id128);                                       /* ncclGetUniqueId: 128 bytes */
int bfhipCommInitRank(const void *id128, int nranks, int rank, int device, BfhipComm **out);
void bfhipCommDestroy(BfhipComm **comm);
int bfhipShardedCreate(BfhipOperator *op, BfhipComm *comm, const BfhipShardSpec *spec, uint32_t maxRhs, BfhipSharded **out);
int bfhipShardedApplyDevice(BfhipSharded *sh, const void *dX, size_t nrhs, void *dY, void *stream);
/* hipEvent times of the most recent apply: stage kernels, and collective (+ reordering).  Synchronizes. */
int bfhipShardedLastTimes(BfhipSharded *sh, double *localMs, double *collectiveMs);
/* The three events behind bfhipShardedLastTimes cost ~17 us of stream time per apply (they are what keeps the stage
 * kernels, the collective and the next apply apart): on by default, turn them off for production loops. */
int bfhipShardedSetTiming(BfhipSharded *sh, int enabled);
/* The adjoint of the same step: dZ[numCols x nrhs] = A^T dV[numRowsGlobal x nrhs] on every rank (v replicated in, the FULL z
 * out), what bfMatRmulVec / cov_matvec need at sizes that only fit sharded (reference src/mat_product.c:312-345,
 * src/mat_block_dense.c:696-758).  Rank r holds A_r -- a set of rows (ROWS) or of blocks (BLOCKS) of A -- so it applies A_r^T
 * to ITS entries of v (its segments, gathered into the order its operator produces them) and gets a full-length partial
 * z_r; the partials add up: ONE in-place ncclAllReduce on dZ.  The local operator needs BFHIP_FLAG_ADJOINT (a row-range
 * shard's adjoint plan is the transposed task list pruned by reachability from its rows; BFHIP_FLAG_ADJOINT_PACKED falls
 * back to that shared-leaf plan on a shard).  Sum order is RCCL's: equal to the one-GPU A^T v to rounding. */
int bfhipShardedApplyTransposeDevice(BfhipSharded *sh, const void *dV, size_t nrhs, void *dZ, void *stream);
/* cov_matvec (examples/covariance/lbo_cov.c:48-60) over a sharded real operator, one call: z = P A G G A^T P' v with the
 * sharded adjoint (all-reduce) and the sharded forward step (all-gather) above; arguments as bfhipCovMatvecDevice, every
 * vector replicated on every rank.  Scratch is allocated in bfhipShardedCreate (real operators with an adjoint plan). */
int bfhipShardedCovMatvecDevice(BfhipSharded *sh, const void *dGammaLam, const uint64_t *dRowPerm, const uint64_t *dRevRowPerm,
                                const void *dV, void *dZ, void *stream);
/* bfSolveGMRES (src/linalg.c:47-317) with the sharded step as the matvec: SURVEY 8(f) row 1, "with multi-GPU the allgathered
 * iterate is already replicated".  Every rank calls it with the same (replicated) dB / dX0 and receives the same dX: the
 * Krylov recurrences of bfhipSolveGMRESOptsDevice run redundantly on every GPU, the only communication of an iteration is
 * the step's one collective, and every rank takes the same decisions (they are computed from identical data).  `opt` as
 * for bfhipSolveGMRESOptsDevice; a preconditioner, if any, is a whole (unsharded) operator on this rank's GPU.  Complex
 * square operators. */
struct BfhipGmresOptions;
int bfhipShardedSolveGMRESDevice(BfhipSharded *sh, const struct BfhipGmresOptions *opt, const void *dB, size_t nrhs, const void *dX0,
                                 size_t *numIter, double *residual, void *dX, void *stream);
size_t bfhipShardedGetNumRows(const BfhipSharded *sh);      /* rows of the whole operator */
size_t bfhipShardedGetNumCols(const BfhipSharded *sh);
/* The vtable shim of bfhipMatNew over a SHARDED operator: every rank's host calls bfMatMul / bfMatMulVec / bfMatRmulVec /
 * bfMatTranspose on its copy with the same (replicated) right-hand side and gets the full result (host vectors are staged
 * through device buffers allocated in bfhipShardedCreate).  Delete releases the shim and, if `ownsSharded`, the sharded
 * object (never the operator or the communicator). */
void *bfhipShardedMatNew(BfhipSharded *sh, int ownsSharded);
/* Neither the operator nor the communicator is released.  Order: free the sharded objects of a communicator before
 * bfhipCommDestroy (a sharded object keeps a pointer to it for its steps; freeing itself does not touch it). */
void bfhipShardedFree(BfhipSharded **sh);
/* Failure semantics of a step: arguments are checked and every allocation is made in bfhipShardedCreate (the vector
 * arena for maxRhs included), so a step cannot fail on one rank for a reason the others do not share.  If a launch
 * still fails locally with more than one rank, the communicator is aborted (ncclCommAbort: the peers' collective
 * returns an error instead of hanging), the step returns non-zero and later steps on that communicator are refused. */

/* Left-preconditioned form: the reference's M argument (src/linalg.c:47-49,90-97,131,159).  `solveM` is a device
 * operator that applies what bfMatSolve(M, .) computes, i.e. the action of M^{-1} (n x n, same device and dtype
 * as `op`); NULL = no preconditioner.  As in the reference the residual is then the preconditioned one. */
int bfhipSolveGMRESPrecondDevice(BfhipOperator *op, BfhipOperator *solveM, const void *dB, size_t nrhs, const void *dX0, double tol,
                                 size_t maxNumIter, size_t *numIter, double *residual, void *dX, void *stream);

/* The same solver with its choices per call.  Orthogonalisation: the reference runs modified Gram-Schmidt one basis
 * vector at a time (src/linalg.c:174-184); CGS2 (classical Gram-Schmidt, two batched passes: 7 launches per iteration
 * whatever j is) is equally stable, 20 % faster on the device and may differ from the reference by one iteration;
 * MGS reproduces the reference's H and iteration count.  DEFAULT = CGS2 unless the environment variable
 * BFHIP_GMRES_MGS=1 asks for MGS (an override for hosts that cannot pass options; an explicit choice here wins). */
enum { BFHIP_GMRES_ORTH_DEFAULT = 0, BFHIP_GMRES_ORTH_CGS2 = 1, BFHIP_GMRES_ORTH_MGS = 2 };
typedef struct BfhipGmresOptions {
  uint32_t structSize;          /* = sizeof(BfhipGmresOptions) */
  uint32_t orthogonalization;   /* BFHIP_GMRES_ORTH_* */
  double tol;
  size_t maxNumIter;
  BfhipOperator *solveM;        /* left preconditioner (action of M^{-1}; complex128, n x n, same device) or NULL */
} BfhipGmresOptions;
int bfhipSolveGMRESOptsDevice(BfhipOperator *op, const BfhipGmresOptions *opt, const void *dB, size_t nrhs, const void *dX0,
                              size_t *numIter, double *residual, void *dX, void *stream);

/* ---- introspection ------------------------------------------------------- */
int bfhipGetStats(const BfhipOperator *op, BfhipStats *stats);
size_t bfhipGetNumRows(const BfhipOperator *op);
size_t bfhipGetNumCols(const BfhipOperator *op);
/* same meaning as bfMatNumBytes on the original graph: leaf payload bytes
 * (mat_block_coo.c:238-258, mat_dense_complex.c:452-455). */
size_t bfhipNumBytes(const BfhipOperator *op);

/* Whether applies of one right-hand side run as one dependency-driven launch (see BFHIP_FLAG_FLOW), and whether
 * one of its waits ever gave up (0 by construction; checked by the test-suite after every flow test). */
int bfhipFlowStatus(BfhipOperator *op, uint32_t *enabled, uint32_t *waitGaveUp);

/* With BFHIP_FLAG_PROFILE: per-stage accumulated kernel time (ms) and launch
 * count since the last reset, measured with hipEvents on the apply stream.
 * `ms`/`launches`/`bytes` are [numStages] arrays (any may be NULL); `bytes`
 * receives the algorithmic bytes one launch of that stage moves for the nrhs
 * of the last apply.  Synchronizes the stream.  An operator whose applies run as ONE launch (bfhipFlowStatus)
 * reports that launch under stage 0 -- time, launch count and the bytes of all stages -- and zeros elsewhere. */
int bfhipGetStageProfile(BfhipOperator *op, double *ms, uint64_t *launches, uint64_t *bytes, int reset);
/* Bracket only one apply in `every` with events (default 1: all of them).  An event pair per launch costs ~4 us of
 * stream time on this hardware (launch gaps of 10 us instead of 6, measured with rocprofv3 --kernel-trace): 2.5 % of
 * an N = 65536 apply, 0.3 % of the headline one.  The sampled launches are timed exactly as before. */
int bfhipSetProfileSampling(BfhipOperator *op, uint32_t every);

/* ---- covariance products: the caller of the real path ------------------------ */
/* examples/covariance/lbo_cov.c:36-60 wraps the streamed butterfly Phi (N x J) in two products, both called in loops:
 *   sample_z:    z = P Phi GammaLam w                         (MulVec, MulVec, bfVecPermute)
 *   cov_matvec:  z = P Phi GammaLam GammaLam Phi^T P' v       (bfVecPermute, RmulVec, MulVec x3, bfVecPermute)
 * with GammaLam a BfMatDiagReal (its J diagonal entries are passed here) and P, P' = bfVecRealPermute with rowPerm /
 * revRowPerm, which SCATTERS: out[perm[i]] = in[i] (src/vec_real.c:312-329).  Through the vtable shim every step is a
 * host vector (two PCIe round trips per product); these entries keep everything on the device, on `stream`:
 * permute, A^T, scale, A, permute.  dGammaLam [numCols] in the operator's element type (NULL: identity), perms
 * uint64 [numRows] (= BfPerm.index; NULL: identity), dW [numCols], dV and dZ [numRows].  Real operators only
 * (TYPE_ERROR otherwise); bfhipCovMatvecDevice needs BFHIP_FLAG_ADJOINT. */
int bfhipCovSampleDevice(BfhipOperator *op, const void *dGammaLam, const uint64_t *dRowPerm, const void *dW, void *dZ, void *stream);
int bfhipCovMatvecDevice(BfhipOperator *op, const void *dGammaLam, const uint64_t *dRowPerm, const uint64_t *dRevRowPerm,
                         const void *dV, void *dZ, void *stream);

/* ---- plan inspection (no device needed) ---------------------------------- */
/* The flattened per-stage layout, as the kernels see it.  Valid only for an
 * operator compiled with BFHIP_FLAG_PLAN_ONLY (the host mirrors are dropped
 * after upload otherwise); pointers stay owned by the operator.  Record
 * layouts: BfDevItem = {u32 pieceBegin, numPieces, outOff, mrFlags},
 * BfDevPiece = {u64 dataOff; u32 inOff, ncols, flags, ld}; in the transposed
 * plan a piece is a forward piece read with lanes on its columns: element
 * (step s, lane j) = arena[dataOff + j*ld + s], ncols = number of steps.
 * mrFlags = rows | flags: 1<<16 the item writes y (else the vector arena);
 * 1<<17 ROWMAJOR (real operands, <= 2 lane granules of rows: the item's dense
 * pieces are stored row by row, element (r, c) = arena[dataOff + r*ld + c],
 * row ends zero-padded: ld is a multiple of the granule, and of 128 bytes for
 * pieces of >= 128 columns, whose rows start on 128-byte lines; one piece per
 * <= 1024-column task);
 * 1<<18 MERGED (column-major dense pieces are one contiguous block of <= 256
 * columns, contracted in one go); 1<<19 SMALL (<= 2 granules of rows, <= 16
 * pieces, < 128 columns: such items are the END of a stage's list and run
 * four to a wavefront in their own launch; their pieces are row-major too);
 * 1<<20 TNARROW (transposed plans: <= 16 columns of a tall leaf; such items
 * are the START of a stage's list and run on the 16-row-lane kernel).  Piece flags: 1 reads x (else the
 * vector arena), 2 identity (no data: adds the input rows), 4 row-major. */
typedef struct BfhipPlanInfo {
  uint32_t structSize, dtype, elemSize, epl, xcap;
  uint32_t reserved;                 /* 1: the adjoint plan is a forward plan of the transposed expression over an arena of its own
                                        (BFHIP_FLAG_ADJOINT_PACKED): its stages are run like forward stages, on bfhipPlanPackArenaT's data */
  uint64_t numRows, numCols, numStages, arenaElems, tempElems;
  uint64_t numStagesT, tempElemsT;   /* transposed plan (0 without BFHIP_FLAG_ADJOINT) */
  uint64_t arenaElemsT;              /* elements of the packed adjoint's own arena (0 otherwise) */
} BfhipPlanInfo;
typedef struct BfhipStageView {
  uint32_t structSize, reserved;
  uint64_t numItems, numPieces, numReduce;
  const void *items;      /* BfDevItem[numItems]  (16 bytes each) */
  const void *pieces;     /* BfDevPiece[numPieces] (24 bytes each) */
  /* filled when structSize covers them: forward complex128 stages -- runs of up to 4 list neighbours that read the same input rows
   * (one workgroup of the 64-RHS kernel each: its wavefronts fetch every X tile once); bundleBegin[numBundles + 1] */
  uint64_t numBundles;
  const uint32_t *bundleBegin;
} BfhipStageView;
typedef struct BfhipReduceView {
  uint32_t structSize, destIsY;
  uint64_t destOff, numRows, numIntervals, numSrc;
  const uint32_t *rowInterval;   /* [numRows] */
  const uint32_t *ivBegin;       /* [numIntervals+1] */
  const int64_t *srcBias;        /* [numSrc] */
} BfhipReduceView;
/* `stage` indices >= numStages address the transposed plan (BFHIP_FLAG_ADJOINT): stage - numStages */
int bfhipPlanGetInfo(const BfhipOperator *op, BfhipPlanInfo *info);
int bfhipPlanGetStage(const BfhipOperator *op, uint64_t stage, BfhipStageView *view);
int bfhipPlanGetReduce(const BfhipOperator *op, uint64_t stage, uint64_t index, BfhipReduceView *view);
/* Write the packed leaf arena (arenaElems elements) to host memory; the
 * descriptor's / graph's leaf values must still be alive.  Synthetic leaves
 * are generated with the host copy of the value stream. */
int bfhipPlanPackArena(const BfhipOperator *op, void *dst);
/* the second arena of a BFHIP_FLAG_ADJOINT_PACKED plan (BfhipPlanInfo.arenaElemsT elements) */
int bfhipPlanPackArenaT(const BfhipOperator *op, void *dst);

/* ---- serialization ------------------------------------------------------- */
/* Write / read a compiled operator (packed leaf arena + per-stage index
 * metadata, both plans if BFHIP_FLAG_ADJOINT) -- the loader the reference's
 * write-only bfMatDump never had (src/mat.c:67-73).  A loaded operator applies
 * bit-identically to the saved one.  Errors: FILE_ERROR (cannot open, bad
 * magic, truncated), MEMORY_ERROR, RUNTIME_ERROR.  `opts` of Load: device,
 * maxRhs, BFHIP_FLAG_PROFILE are honoured. */
int bfhipSave(BfhipOperator *op, const char *path);
int bfhipLoad(const char *path, const BfhipOptions *opts, BfhipOperator **out);

/* ---- lifetime ------------------------------------------------------------ */
void bfhipFree(BfhipOperator **op);

/* ---- reference-vtable shim ----------------------------------------------- */

/* A `BfMat *` whose vtable implements Mul, Rmul, MulVec, RmulVec, Transpose, GetView, GetNumRows,
 * GetNumCols, GetType (-> BF_TYPE_MAT_FUNC), NumBytes and Delete on top of
 * `op`, so that unmodified reference code (bfSolveGMRES src/linalg.c:125,155;
 * cov_matvec examples/covariance/lbo_cov.c:48-60; an enclosing BfMatBlockDense,
 * src/mat_block_dense.c:541-563) can call bfMatMul / bfMatMulVec / bfMatRmulVec
 * on it.  Mul results are allocated through the RHS's own vtable (`EmptyLike`,
 * slot 8) so that the reference's bfMatDelete frees them
 * (mat_dense_complex.c:2164-2187); Rmul (slot 44: bfMatRmul(A, X) = X A for a
 * BfMatDenseComplex X, src/mat.c:195-197, src/mat_product.c:282-310,
 * src/mat_dense_complex.c:1075-1133) applies the adjoint plan to the rows of X
 * (BFHIP_FLAG_ADJOINT) and allocates X A the same way; MulVec / RmulVec results are malloc'd
 * BfVecReal of the operator's row / column count carrying the argument's
 * vtable (rectangular operators are fine).  Transpose (slot 63, bfMatTranspose,
 * src/mat.c:271-273) works in place like bfMatProductTranspose
 * (src/mat_product.c:409-420): the object switches between the forward and the
 * adjoint plan of `op` (BFHIP_FLAG_ADJOINT; without it the slot raises
 * BF_ERROR_NOT_IMPLEMENTED and changes nothing), GetNumRows / GetNumCols answer
 * for A^T, MulVec multiplies by A^T and RmulVec by A (real operators); Mul on a
 * complex operator multiplies by A^H afterwards, as in the reference, whose dense
 * complex leaves transpose by bfMatConjTrans and multiply through CblasConjTrans
 * (src/mat_dense_complex.c:1475-1478, 27-35); twice is the identity.
 * Delete releases the shim and, if `ownsOperator`, the operator; a GetView copy
 * never owns it (and is transposed if its source was). */
void *bfhipMatNew(BfhipOperator *op, int ownsOperator);

/* Failures of the shim's Mul / MulVec / RmulVec / GetView return NULL and, by default, also raise
 * the reference's global error state: `bfSetError(code)` (src/error.c:20-24) is looked up in the
 * host process at run time (libbfhip does not link the reference).  bfSetError asserts on a
 * non-zero code, so in a reference build with assertions this is as fatal as the reference's own
 * BF_DIE() paths; pass 0 to keep failures to NULL + bfhipLastErrorMessage(). */
void bfhipSetErrorForwarding(int on);

/* `MatMulFunc` for the reference's own callback operator
 * (include/bf/mat_func.h:5): bfMatFuncInit(f, m, n, bfhipMatMulFunc, op). */
void *bfhipMatMulFunc(const void *rhsBfMat, void *op);

/* ---- errors -------------------------------------------------------------- */
const char *bfhipErrorString(int code);
/* message of the most recent failure on this thread ("" if none) */
const char *bfhipLastErrorMessage(void);

/* Value of element `idx` of the synthetic value stream for `seed`: uniform in
 * [-1, 1).  Identical on host and device (exact IEEE operations), so a CPU
 * restatement can rebuild the very operand the device synthesized.  A leaf
 * with node id L and m x n row-major elements uses
 *   idx = leafBase(L) + i*n + j  (re) and the same idx with stream bit set (im),
 * scaled by sqrt(3/(2n)) (complex) or sqrt(3/n) (real). */
double bfhipSyntheticValue(uint64_t seed, uint64_t idx, int imag);   /* = bfhip_synth_value, include/bfhip_synth.h */
/* virtual base index of leaf `node` (prefix sum of rows*cols over dense nodes
 * in node order), for a given descriptor */
int bfhipSyntheticLeafBases(const BfhipDesc *desc, uint64_t *bases);

#ifdef __cplusplus
}
#endif
#endif /* BFHIP_H */
