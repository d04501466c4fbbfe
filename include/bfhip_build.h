/* bfhip_build.h -- device builder for fac_helm2 operands.
 *
 * The apply engine (bfhip.h) takes an already-built factorization.  This
 * extension computes the leaf VALUES of a 2-D Helmholtz butterfly on the
 * MI355X, straight into the operator's packed arena, from the block layout
 * (BfhipDesc) plus one recipe per dense leaf.  It replaces the value side of
 *
 *   bfFacHelm2MakeMultilevel / bfFacHelm2Make    src/fac_helm2.c:653-704, 943-1003
 *     makeFirstFactor   :42-160    re-expansion  source points -> proxy circle
 *     makeFactor        :222-401   re-expansion  child circle   -> parent circle
 *     makeLastFactor    :403-509   evaluation    proxy circle   -> target points
 *     dense near field  :888-915   kernel matrix points -> points
 *   bfHelm2GetKernelMatrix (single layer)         src/helm2.c:93-125, 186-215
 *   bfHelm2GetReexpansionMatrix                    src/helm2.c:321-365
 *   bfMatDenseComplexDenseComplexLstSq             src/mat_dense_complex.c:1767-1849
 *   bfCircle2SamplePoints                          src/circle.c:12-35
 *
 * (the structure side -- quadtree, level choice, ranks -- stays on the host:
 * butterfly_amd/helm2_structure.py restates it; a C caller passes the layout
 * of a reference-built operand).  Numerics: the kernel is (i/4) H0^(1)(k r), 0
 * at r == 0; a re-expansion leaf is X = pinv_trunc(Z_equiv) Z_orig with the
 * reference's truncation rule (singular values below max(m,n) eps s_max + eps
 * dropped), computed by a one-sided Jacobi SVD (QR-preconditioned with column pivoting from 65 equivalent sources up)
 * instead of LAPACK zgesvd.  Where the two criteria differ: the QR stage stops on COLUMN NORMS (the largest remaining column
 * of the pivoted factorization below max(m,n) eps x the largest column of Z_equiv -- the threshold under which the Jacobi kernel
 * freezes a column), the reference truncates on SINGULAR VALUES (sigma < max(m,n) eps sigma_max + eps).  The stop is at the noise
 * floor, far below the rule's threshold on these matrices, and the rule itself is then applied to the singular values of what the
 * QR kept; a direction whose singular value lies within a factor ~sqrt(columns) of the threshold can still be dropped by the QR
 * stage where zgesvd would have kept it -- such directions are rounding noise in either computation (BFHIP_JACOBI_QR_MIN=<huge>
 * turns the stage off).  A least-squares matrix that is zero or not finite (QR rank 0) is counted in
 * BfhipBuildStats.notConverged: the build fails unless BFHIP_ALLOW_UNCONVERGED_SVD=1, and the leaf is then exactly zero.
 * Element-wise agreement with the CPU path is therefore at the level of the
 * truncation (the dropped directions), while Z_equiv X, and hence every
 * apply result, agrees to ~1e-12; see tests/test_gpu_build.py.
 */
#ifndef BFHIP_BUILD_H
#define BFHIP_BUILD_H

#include "bfhip.h"

#ifdef __cplusplus
extern "C" {
#endif

enum {
  BFHIP_PTS_TREE = 0,    /* points[first .. first+count) of the (quadtree-ordered) point array */
  BFHIP_PTS_CIRCLE = 1,  /* count points (cx + r cos(2 pi i/count), cy + r sin(2 pi i/count)), src/circle.c:12-35 */
  BFHIP_PTS_TREE_TGT = 2 /* tgtPoints[first .. first+count): a separate target tree (bfFacHelm2MakeMultilevel with
                            srcTree != tgtTree, e.g. examples/multiple_scattering/multiple_scattering_context.c:998) */
};

typedef struct BfhipPointSet {
  uint32_t kind;         /* BFHIP_PTS_* */
  uint32_t count;
  uint64_t first;        /* TREE only */
  double cx, cy, r;      /* CIRCLE only */
} BfhipPointSet;

enum {
  BFHIP_LEAF_KERNEL = 0, /* leaf[i][j] = G(tgt_i, src_j): rows = tgt.count, cols = src.count */
  BFHIP_LEAF_REEXP = 1   /* leaf = lstsq(G(tgt, equiv), G(tgt, src)): rows = equiv.count, cols = src.count;
                            needs tgt.count >= equiv.count */
};

typedef struct BfhipHelm2Recipe {
  uint64_t node;         /* dense leaf of the descriptor this recipe fills */
  uint32_t kind;         /* BFHIP_LEAF_* */
  uint32_t reserved;
  BfhipPointSet src;     /* original sources (columns of the leaf) */
  BfhipPointSet equiv;   /* REEXP: equivalent sources (rows of the leaf) */
  BfhipPointSet tgt;     /* KERNEL: targets (rows); REEXP: check points */
} BfhipHelm2Recipe;

/* layer potentials, numbered as the reference's BfLayerPotential (include/bf/layer_pot.h:27-42) */
enum {
  BFHIP_LAYER_POTENTIAL_SINGLE = 1,                 /* S : G(x,y) = (i/4) H0(k r)                    src/helm2.c:93-125  */
  BFHIP_LAYER_POTENTIAL_PV_DOUBLE = 2,              /* D : (i/4) k H1(k r)/r  n_src.(x_tgt - x_src)  src/helm2.c:173-218 */
  BFHIP_LAYER_POTENTIAL_PV_NORMAL_DERIV_SINGLE = 3, /* S': (i/4) k H1(k r)/r  n_tgt.(x_tgt - x_src)  src/helm2.c:126-171 */
  BFHIP_LAYER_POTENTIAL_COMBINED_FIELD = 5          /* alpha S + beta D                              src/helm2.c:220-279 */
};

typedef struct BfhipHelm2Problem {
  uint32_t structSize;       /* = sizeof(BfhipHelm2Problem) */
  uint32_t layerPot;         /* BFHIP_LAYER_POTENTIAL_*; others: NOT_IMPLEMENTED.  As in the reference, for S' only
                                KERNEL leaves (evaluation factor, dense near field) use S' and re-expansions use
                                the proxy potential S; D and the combined field are their own proxy potential
                                (BF_PROXY_LAYER_POT, layer_pot.h:63-69, fac_helm2.c:59,231).  Source normals of a
                                circle point set are its radial unit vectors (bfCircle2SampleUnitNormals,
                                src/circle.c:36-58). */
  double wavenumber;
  const double *points;      /* [2 * numPoints] host, (x, y) pairs in quadtree order */
  uint64_t numPoints;
  const BfhipHelm2Recipe *recipes;
  uint64_t numRecipes;
  uint64_t workspaceBytes;   /* device scratch for one batch of leaves; 0 -> half of the free device memory */
  const double *normals;     /* [2 * numPoints] unit normals at the points (S': target, D / combined: source
                                normals); may be NULL for S */
  /* system-matrix decorations folded into the values (reference: bfMatScaleCols + bfMatAddInplace of c I,
   * examples/simple/helm2_bie.c:109-121): the built operator is  selfValue * I + K * diag(colWeights) */
  const double *colWeights;  /* [numPoints] or NULL: columns whose sources are points j are scaled by colWeights[j] */
  double selfValue[2];       /* (re, im) of the entries with target point == source point (the kernel has 0 there) */
  /* Kapur-Rokhlin correction of the punctured trapezoid rule on a closed curve
   * (bfHelm2ApplyKrCorrectionTree -> bfQuadKrApplyCorrectionTree, src/quadrature.c:126-199;
   * helm2_bie.c:113): entry (i, j) of the layer potential is multiplied by 1 + w_KR[d-1] when the
   * points are d = 1..krOrder apart along the curve (cyclic distance of their ORIGINAL indices).
   * Folded into the dense near-field leaves; the build fails if such a pair lies in a butterflied
   * block.  krOrder: 0 (none), 2, 6 or 10; origIndex[t] = original index of tree position t. */
  const uint64_t *origIndex; /* [numPoints] or NULL (needed iff krOrder != 0) */
  uint32_t krOrder;
  uint32_t reserved;
  double alpha[2], beta[2];  /* COMBINED_FIELD: complex coefficients (BfHelm2.alpha, .beta, include/bf/helm2.h:13-14) */
  /* a separate target tree: rows of the operator follow tgtPoints (quadtree order of THAT tree).
   * NULL: targets are `points` (square operator).  selfValue / KR apply to square operators only. */
  const double *tgtPoints;   /* [2 * numTgtPoints] or NULL */
  uint64_t numTgtPoints;
  const double *tgtNormals;  /* [2 * numTgtPoints]: S' with a separate target tree; else may be NULL */
} BfhipHelm2Problem;

typedef struct BfhipBuildStats {
  uint32_t structSize;
  uint32_t numBatches;
  uint64_t kernelLeaves, reexpLeaves;
  uint64_t kernelEvals;      /* Hankel evaluations */
  uint64_t maxSweeps;        /* largest Jacobi sweep count over all problems */
  uint64_t notConverged;     /* problems that hit the sweep cap */
  uint64_t truncated;        /* singular values dropped, total */
  uint64_t sumSweeps;        /* Jacobi sweeps summed over all problems */
  double seconds;            /* wall time of the value build (device work + orchestration) */
  uint64_t qrProblems;       /* least-squares problems that went through the QR preconditioner (>= 65 equivalent sources) */
  uint64_t qrColumns;        /* their columns, total ... */
  uint64_t qrRank;           /* ... and how many of those the Jacobi kernel still had to orthogonalise */
} BfhipBuildStats;

/* Compile `desc` (all dense leaves must have leafData == NULL and a recipe)
 * and compute the leaf values on the device.  `opts` as bfhipCompileDesc
 * (sharding keeps only the recipes whose leaves survive).  `stats` may be NULL. */
int bfhipBuildHelm2(const BfhipDesc *desc, const BfhipHelm2Problem *prob, const BfhipOptions *opts,
                    BfhipOperator **out, BfhipBuildStats *stats);

/* ---- layout: points -> block structure + recipes (host only) ------------------------------
 * What bfFacHelm2MakeMultilevel(helm, quadtree, quadtree) would lay out for `points` (quadtree with
 * leaf size 1, src tree = tgt tree: src/fac_helm2.c:943-1002 and callees, see bfhip_layout.c): the
 * flat descriptor bfhipCompileDesc / bfhipBuildHelm2 take, one recipe per dense leaf, and the
 * quadtree permutation.  All returned pointers belong to the layout. */
typedef struct BfhipHelm2Layout BfhipHelm2Layout;
int bfhipHelm2LayoutCreate(const double *points, uint64_t numPoints, double wavenumber, BfhipHelm2Layout **out);
/* src tree != tgt tree: rows follow the quadtree on tgtPoints, columns the one on points */
int bfhipHelm2LayoutCreate2(const double *points, uint64_t numPoints, const double *tgtPoints, uint64_t numTgtPoints,
                            double wavenumber, BfhipHelm2Layout **out);
const uint64_t *bfhipHelm2LayoutGetTgtPerm(const BfhipHelm2Layout *layout);       /* NULL for a single tree */
const double *bfhipHelm2LayoutGetTgtTreePoints(const BfhipHelm2Layout *layout);   /* NULL for a single tree */
/* bfFacHelm2MakeSingleLevel (src/fac_helm2.c:706-729; examples/simple/bf_one_block.c:162): the butterfly of
 * ONE (source node, target node) pair; nodes named by their paths from the root (child positions among
 * the non-empty children, quadrant order; same depth).  Maps the source node's points to the target
 * node's points; the recipes address the whole tree-ordered point array. */
int bfhipHelm2LayoutCreateSingle(const double *points, uint64_t numPoints, double wavenumber, const uint32_t *srcPath, uint32_t srcDepth,
                                 const uint32_t *tgtPath, uint32_t tgtDepth, BfhipHelm2Layout **out);
const BfhipDesc *bfhipHelm2LayoutGetDesc(const BfhipHelm2Layout *layout);
const BfhipHelm2Recipe *bfhipHelm2LayoutGetRecipes(const BfhipHelm2Layout *layout, uint64_t *count);
const uint64_t *bfhipHelm2LayoutGetPerm(const BfhipHelm2Layout *layout);        /* perm[t] = original index of tree position t */
const double *bfhipHelm2LayoutGetTreePoints(const BfhipHelm2Layout *layout);    /* [2 * numPoints], tree order */
void bfhipHelm2LayoutFree(BfhipHelm2Layout **layout);

/* ---- layout of a streamed real butterfly (BASELINE configs[4]; host only) -------------------------------
 * The block structure `bfFacSpanGetMat(bfFacStreamerGetFacSpan(fs))` hands to examples/covariance/lbo_cov.c:188-189
 * -- a 1 x numFacs BlockDense row of products [Psi, W0, W1, ...] (src/fac_span.c:126-155, src/fac.c:53-75) -- laid out by
 * the streamer's own recursion (octree rows with leaf size 1, complete binary frequency tree in post order,
 * bfFacStreamerFeed, mergeAndSplit, epsilon-rank cut: see bfhip_streamer_layout.c) WITHOUT values: the truncated SVD of a
 * block is answered by the rank model
 *     rank(f, [w0, w1)) = (alpha sqrt(f) w1 + delta)^2 - max(alpha sqrt(f) w0 - delta, 0)^2,  f = rows / numPoints,
 * clipped by the block and by the band's column count (the local Weyl count of a surface patch against a frequency band;
 * butterfly_amd/streamer_structure.py: LboRankModel, fitted to SVD-driven structures).  This is how the benchmark operand
 * of N = 1M rows x 65536 columns gets the shapes the reference would build; parity operands with real values come from
 * the Python restatement + numpy SVDs (oracle/streamer_values.py), which this layout equals array for array under the
 * same rank answers (tests/test_streamer_layout_cpu.py). */
typedef struct BfhipStreamerSpec {
  uint32_t structSize;          /* = sizeof(BfhipStreamerSpec) */
  uint32_t colDepth;            /* depth of the frequency tree: 2^colDepth leaf bands over [0, wmax) */
  double wmax;                  /* upper end of the frequency range */
  const uint64_t *bandColumns;  /* [2^colDepth] columns fed per leaf band, left to right (0 allowed: still a feed) */
  uint64_t minNumRows, minNumCols;   /* bfFacSpec.minNumRows / minNumCols (0 -> 20, lbo_cov.c:126-127) */
  uint64_t maxCols;             /* stop feeding once this many columns are in (lbo_cov.c:139-143); 0 = all bands */
  double alpha, delta;          /* rank model (0 -> 1.75, 3.0: the tol = 1e-3 fit) */
} BfhipStreamerSpec;
typedef struct BfhipStreamerStats {
  uint32_t structSize, maxNest;
  uint64_t numRows, numCols, numFacs, numW, rowNodes;      /* numW / rowNodes: of the last partial factorization */
  uint64_t product, blockCoo, blockDense, blockDiag, denseReal, identity, leafBytes;   /* the graph, node by node (fp64 bytes) */
  uint64_t svds, merges, feeds, octreeDepth;
} BfhipStreamerStats;
typedef struct BfhipStreamerLayout BfhipStreamerLayout;
/* points: [3 * numPoints] in the caller's order; rows of the operand follow the octree (GetPerm: perm[t] = original index) */
int bfhipStreamerLayoutCreate(const double *points, uint64_t numPoints, const BfhipStreamerSpec *spec, BfhipStreamerLayout **out);
const BfhipDesc *bfhipStreamerLayoutGetDesc(const BfhipStreamerLayout *layout);      /* dtype BFHIP_F64, no leaf values */
const uint64_t *bfhipStreamerLayoutGetPerm(const BfhipStreamerLayout *layout);
int bfhipStreamerLayoutGetStats(const BfhipStreamerLayout *layout, BfhipStreamerStats *stats);
void bfhipStreamerLayoutFree(BfhipStreamerLayout **layout);
/* depth of the octree alone: lbo_cov.c:97-98 derives the frequency-tree depth from it (row-tree depth - 3) */
int bfhipStreamerOctreeDepth(const double *points, uint64_t numPoints, uint32_t *depth);

/* Points -> device operator in one call (layout + bfhipBuildHelm2): the device counterpart of
 * bfFacHelm2MakeMultilevel followed by the decorations of examples/simple/helm2_bie.c.  `points`,
 * `normals` (may be NULL for S), `colWeights` (may be NULL) are in the caller's ORIGINAL order;
 * `params` supplies layerPot, wavenumber, selfValue, krOrder, alpha, beta, workspaceBytes (its
 * pointer fields are ignored).  The operator acts on vectors in quadtree order, as the
 * reference's does; permOut[t] (may be NULL) = original index of position t. */
int bfhipFacHelm2MakeMultilevel(const double *points, const double *normals, const double *colWeights, uint64_t numPoints,
                                const BfhipHelm2Problem *params, const BfhipOptions *opts, BfhipOperator **out, uint64_t *permOut,
                                BfhipBuildStats *stats);

/* The same with a separate target tree (srcTree != tgtTree): tgtPoints / tgtNormals (may be NULL unless
 * S') in the caller's order; rows of the operator follow the target quadtree, tgtPermOut[t] (may be
 * NULL) = original index of target position t.  tgtPoints == NULL: as above. */
int bfhipFacHelm2MakeMultilevel2(const double *points, const double *normals, const double *colWeights, uint64_t numPoints,
                                 const double *tgtPoints, const double *tgtNormals, uint64_t numTgtPoints,
                                 const BfhipHelm2Problem *params, const BfhipOptions *opts, BfhipOperator **out, uint64_t *permOut,
                                 uint64_t *tgtPermOut, BfhipBuildStats *stats);

/* One leaf, computed on the device and returned to the host row-major
 * (rows x cols complex128) -- unit-level parity checks of the builder. */
int bfhipHelm2BuildLeaf(const BfhipHelm2Problem *prob, uint64_t recipeIndex, int device, void *out);

/* y = G x with the N x N single-layer kernel matrix evaluated on the fly
 * (never stored): the reference examples' acceptance check at sizes where
 * the dense matrix does not fit (examples/simple/bf_all_blocks.c:132-153).
 * Uses points, wavenumber, layerPot, normals, colWeights and selfValue of `prob`
 * (recipes are ignored).  dX: device, numPoints complex128; dY: numTgtPoints if tgtPoints is
 * given, else numPoints. */
int bfhipHelm2DenseApplyDevice(const BfhipHelm2Problem *prob, int device, const void *dX, void *dY, void *stream);
int bfhipHelm2DenseApply(const BfhipHelm2Problem *prob, int device, const void *X, void *Y);

#ifdef __cplusplus
}
#endif
#endif /* BFHIP_BUILD_H */
