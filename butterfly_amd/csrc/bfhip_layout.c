/* bfhip_layout.c -- block layout of a fac_helm2 multilevel butterfly, on the host, in C.
 *
 * From the points and the wavenumber alone this derives every block shape the
 * reference's builder produces and one value recipe per dense leaf, i.e. the
 * input of bfhipBuildHelm2 -- so that a caller can go from points to a device
 * operator without the reference's CPU build.  It follows the same logic as
 * butterfly_amd/helm2_structure.py (which the tests hold it to, array for array):
 *
 *   quadtree, leaf size 1, square root box       reference src/quadtree_node.c:17,123-282, src/bbox.c:18-30
 *   bounding circles, separation test             src/quadtree_node.c:321-330,393-401
 *   rank rule p = ceil(k r1 r2 / d - log10 eps)   src/helm2.c:14-36
 *   level selection                               src/fac_helm2.c:510-530, 551-651
 *   first / inner / last factor layout            src/fac_helm2.c:42-160, 222-401, 403-509
 *   dense threshold, recursion                    src/fac_helm2.c:20, 860-941, 943-1002
 *
 * (The quadrant partition is stable; the reference's in-place sift is not for
 * non-members, so point order inside a node can differ from the reference's
 * while the block layout does not.)
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "bfhip_internal.h"
#include "../../include/bfhip_abi.h"
#include "../../include/bfhip_build.h"

#define MAX_DENSE_MATRIX_SIZE (128 * 128)   /* src/fac_helm2.c:20 */
#define EPS_MACH 2.220446049250313e-16
#define MAX_DEPTH 200                       /* coincident points would subdivide for ever */

typedef struct QN {
  double xmin, ymin, xmax, ymax, cx, cy, r;
  uint64_t i0, i1;
  uint32_t depth, nch;
  uint32_t ch[4];
  uint32_t tree;            /* 0: the source tree (or the only one), 1: a separate target tree */
  /* levels of the subtree, built on demand: nodes of level d are lv[lvBegin[d] .. lvBegin[d+1]) */
  uint32_t *lv, *lvBegin;
  uint32_t nlv;
} QN;

struct BfhipHelm2Layout {
  /* quadtree */
  QN *nodes;
  uint32_t numNodes, capNodes;
  uint64_t n;
  uint64_t *perm;
  double *treePoints;
  /* separate target tree (NULL / 0 when src tree == tgt tree) */
  uint64_t m;
  uint64_t *tperm;
  double *ttreePoints;
  /* descriptor under construction */
  uint8_t *kind, *blockKind;
  uint64_t *rows, *cols, *childBegin;
  uint64_t numDesc, capDesc;
  uint64_t *childNode, *childRow0, *childCol0;
  uint64_t numChildren, capChildren;
  uint64_t *topRowBlock;
  BfhipHelm2Recipe *recipes;
  uint64_t numRecipes, capRecipes;
  BfhipDesc desc;
  double k;
  int oom;
};

/* ---- growing arrays ---------------------------------------------------------------- */
static void *grow(void *p, uint64_t *cap, uint64_t need, size_t elem, int *oom) {
  if (need <= *cap) return p;
  uint64_t nc = *cap ? *cap * 2 : 1024;
  while (nc < need) nc *= 2;
  void *q = realloc(p, (size_t)nc * elem);
  if (!q) { *oom = 1; return p; }
  *cap = nc;
  return q;
}

typedef struct Child { uint64_t node, r0, c0; } Child;

static uint64_t descAdd(BfhipHelm2Layout *L, uint8_t kind, uint64_t rows, uint64_t cols, Child const *ch, uint64_t nch, uint8_t blockKind) {
  uint64_t cap = L->capDesc;
  if (L->numDesc + 2 > cap) {
    uint64_t c1 = cap, c2 = cap, c3 = cap, c4 = cap, c5 = cap;
    L->kind = grow(L->kind, &c1, L->numDesc + 2, 1, &L->oom);
    L->blockKind = grow(L->blockKind, &c2, L->numDesc + 2, 1, &L->oom);
    L->rows = grow(L->rows, &c3, L->numDesc + 2, 8, &L->oom);
    L->cols = grow(L->cols, &c4, L->numDesc + 2, 8, &L->oom);
    L->childBegin = grow(L->childBegin, &c5, L->numDesc + 2, 8, &L->oom);
    if (L->oom) return 0;
    L->capDesc = c1;
  }
  if (L->numChildren + nch > L->capChildren) {
    uint64_t c1 = L->capChildren, c2 = L->capChildren, c3 = L->capChildren;
    L->childNode = grow(L->childNode, &c1, L->numChildren + nch, 8, &L->oom);
    L->childRow0 = grow(L->childRow0, &c2, L->numChildren + nch, 8, &L->oom);
    L->childCol0 = grow(L->childCol0, &c3, L->numChildren + nch, 8, &L->oom);
    if (L->oom) return 0;
    L->capChildren = c1;
  }
  uint64_t const id = L->numDesc++;
  L->kind[id] = kind; L->blockKind[id] = blockKind; L->rows[id] = rows; L->cols[id] = cols;
  L->childBegin[id] = L->numChildren;
  for (uint64_t i = 0; i < nch; ++i) {
    L->childNode[L->numChildren] = ch[i].node; L->childRow0[L->numChildren] = ch[i].r0; L->childCol0[L->numChildren] = ch[i].c0;
    L->numChildren++;
  }
  L->childBegin[id + 1] = L->numChildren;
  return id;
}

static BfhipPointSet nodePts(QN const *nd) {
  BfhipPointSet p;
  memset(&p, 0, sizeof p);
  p.kind = nd->tree ? BFHIP_PTS_TREE_TGT : BFHIP_PTS_TREE; p.first = nd->i0; p.count = (uint32_t)(nd->i1 - nd->i0);
  return p;
}
static BfhipPointSet circPts(QN const *nd, uint64_t count) {
  BfhipPointSet p;
  memset(&p, 0, sizeof p);
  p.kind = BFHIP_PTS_CIRCLE; p.count = (uint32_t)count; p.cx = nd->cx; p.cy = nd->cy; p.r = nd->r;
  return p;
}

static void addRecipe(BfhipHelm2Layout *L, uint64_t node, uint32_t kind, BfhipPointSet src, BfhipPointSet equiv, BfhipPointSet tgt) {
  L->recipes = grow(L->recipes, &L->capRecipes, L->numRecipes + 1, sizeof(BfhipHelm2Recipe), &L->oom);
  if (L->oom) return;
  BfhipHelm2Recipe *r = &L->recipes[L->numRecipes++];
  memset(r, 0, sizeof *r);
  r->node = node; r->kind = kind; r->src = src; r->equiv = equiv; r->tgt = tgt;
}

static uint64_t addKernelLeaf(BfhipHelm2Layout *L, uint64_t rows, uint64_t cols, BfhipPointSet src, BfhipPointSet tgt) {
  uint64_t const id = descAdd(L, BFHIP_NODE_DENSE, rows, cols, NULL, 0, 0);
  BfhipPointSet none;
  memset(&none, 0, sizeof none);
  addRecipe(L, id, BFHIP_LEAF_KERNEL, src, none, tgt);
  return id;
}

/* ---- quadtree ---------------------------------------------------------------------- */
static uint32_t newNode(BfhipHelm2Layout *L, double xmin, double ymin, double xmax, double ymax, uint64_t i0, uint64_t i1, uint32_t depth,
                        uint32_t tree) {
  if (L->numNodes == L->capNodes) {
    uint64_t cap = L->capNodes;
    L->nodes = grow(L->nodes, &cap, (uint64_t)L->numNodes + 1, sizeof(QN), &L->oom);
    if (L->oom) return 0;
    L->capNodes = (uint32_t)cap;
  }
  QN *nd = &L->nodes[L->numNodes];
  memset(nd, 0, sizeof *nd);
  nd->xmin = xmin; nd->ymin = ymin; nd->xmax = xmax; nd->ymax = ymax; nd->i0 = i0; nd->i1 = i1; nd->depth = depth; nd->tree = tree;
  nd->r = hypot(xmax - xmin, ymax - ymin) / 2;         /* bounding circle: src/quadtree_node.c:321-330 */
  nd->cx = (xmin + xmax) / 2;
  nd->cy = (ymin + ymax) / 2;
  return L->numNodes++;
}

static int buildQuadtree(BfhipHelm2Layout *L, double const *pts, uint64_t n, uint64_t *perm, uint32_t tree, uint32_t *rootOut) {
  double xmin = pts[0], xmax = pts[0], ymin = pts[1], ymax = pts[1];
  for (uint64_t i = 1; i < n; ++i) {
    double const x = pts[2 * i], y = pts[2 * i + 1];
    if (x < xmin) xmin = x;
    if (x > xmax) xmax = x;
    if (y < ymin) ymin = y;
    if (y > ymax) ymax = y;
  }
  /* bfBbox2RescaleToSquare, src/bbox.c:18-30 */
  double const w = xmax - xmin, h = ymax - ymin;
  if (w > h) {
    double const c = (ymin + ymax) / 2, lo = w * (ymin - c) / h + c, hi = w * (ymax - c) / h + c;
    ymin = lo; ymax = hi;
  } else {
    double const c = (xmin + xmax) / 2, lo = h * (xmin - c) / w + c, hi = h * (xmax - c) / w + c;
    xmin = lo; xmax = hi;
  }
  for (uint64_t i = 0; i < n; ++i) perm[i] = i;
  uint64_t *tmp = malloc((size_t)n * 8);
  uint8_t *quad = malloc((size_t)n);
  uint32_t *stack = NULL;
  uint64_t sp = 0, capStack = 0;
  int rc = 0;
  if (!tmp || !quad) { rc = bfhipFail(BFABI_ERROR_MEMORY_ERROR, "host OOM (quadtree)"); goto done; }
  uint32_t const root = newNode(L, xmin, ymin, xmax, ymax, 0, n, 0, tree);
  *rootOut = root;
  stack = grow(stack, &capStack, 1, 4, &L->oom);
  if (L->oom) goto done;
  stack[sp++] = root;
  while (sp && !L->oom) {
    uint32_t const id = stack[--sp];
    QN nd = L->nodes[id];                                /* copy: the array may move */
    if (nd.depth >= MAX_DEPTH) { rc = bfhipFail(BFABI_ERROR_INVALID_ARGUMENTS, "quadtree deeper than %d levels: coincident points?", MAX_DEPTH); goto done; }
    double const sx = nd.cx, sy = nd.cy;                 /* split = box centre (src/quadtree_node.c:237) */
    uint64_t cnt[4] = {0, 0, 0, 0}, off[5];
    for (uint64_t i = nd.i0; i < nd.i1; ++i) {
      double const x = pts[2 * perm[i]], y = pts[2 * perm[i] + 1];
      uint8_t const q = x <= sx ? (y <= sy ? 0 : 1) : (y <= sy ? 2 : 3);
      quad[i] = q;
      cnt[q]++;
    }
    off[0] = nd.i0;
    for (int q = 0; q < 4; ++q) off[q + 1] = off[q] + cnt[q];
    uint64_t cur[4] = {off[0], off[1], off[2], off[3]};
    for (uint64_t i = nd.i0; i < nd.i1; ++i) tmp[cur[quad[i]]++] = perm[i];         /* stable */
    memcpy(perm + nd.i0, tmp + nd.i0, (size_t)(nd.i1 - nd.i0) * 8);
    double const box[4][4] = {{nd.xmin, nd.ymin, sx, sy}, {nd.xmin, sy, sx, nd.ymax}, {sx, nd.ymin, nd.xmax, sy}, {sx, sy, nd.xmax, nd.ymax}};
    for (int q = 0; q < 4; ++q) {
      if (off[q + 1] == off[q]) continue;
      uint32_t const c = newNode(L, box[q][0], box[q][1], box[q][2], box[q][3], off[q], off[q + 1], nd.depth + 1, tree);
      if (L->oom) break;
      QN *par = &L->nodes[id];
      par->ch[par->nch++] = c;
      if (off[q + 1] - off[q] > 1) {                     /* leaf size threshold 1 (src/quadtree_node.c:17) */
        stack = grow(stack, &capStack, sp + 1, 4, &L->oom);
        if (L->oom) break;
        stack[sp++] = c;
      }
    }
  }
done:
  free(tmp); free(quad); free(stack);
  if (!rc && L->oom) rc = bfhipFail(BFABI_ERROR_MEMORY_ERROR, "host OOM (quadtree)");
  return rc;
}

/* BFS levels of the subtree of `id` (src/tree_level_iter.c:10-44), cached on the node */
static int ensureLevels(BfhipHelm2Layout *L, uint32_t id) {
  if (L->nodes[id].lv) return 0;
  uint64_t capN = 0, capB = 0, nn = 0, nb = 0;
  uint32_t *lv = NULL, *lb = NULL;
  lv = grow(lv, &capN, 1, 4, &L->oom);
  lb = grow(lb, &capB, 2, 4, &L->oom);
  if (L->oom) goto fail;
  lv[nn++] = id; lb[nb++] = 0;
  for (;;) {
    uint64_t const b = lb[nb - 1], e = nn;
    for (uint64_t i = b; i < e; ++i) {
      QN const *nd = &L->nodes[lv[i]];
      for (uint32_t c = 0; c < nd->nch; ++c) {
        lv = grow(lv, &capN, nn + 1, 4, &L->oom);
        if (L->oom) goto fail;
        lv[nn++] = nd->ch[c];
      }
    }
    lb = grow(lb, &capB, nb + 2, 4, &L->oom);
    if (L->oom) goto fail;
    lb[nb++] = (uint32_t)e;
    if (nn == e) break;                                  /* the level after the deepest one is empty */
  }
  L->nodes[id].lv = lv; L->nodes[id].lvBegin = lb; L->nodes[id].nlv = (uint32_t)(nb - 1);
  return 0;
fail:
  free(lv); free(lb);
  return bfhipFail(BFABI_ERROR_MEMORY_ERROR, "host OOM (levels)");
}

static uint32_t const *levelNodes(BfhipHelm2Layout const *L, uint32_t id, uint32_t d, uint32_t *count) {
  QN const *nd = &L->nodes[id];
  *count = nd->lvBegin[d + 1] - nd->lvBegin[d];
  return nd->lv + nd->lvBegin[d];
}
static int levelInternal(BfhipHelm2Layout const *L, uint32_t id, uint32_t d) {
  uint32_t cnt;
  uint32_t const *ns = levelNodes(L, id, d, &cnt);
  for (uint32_t i = 0; i < cnt; ++i)
    if (L->nodes[ns[i]].nch == 0) return 0;
  return 1;
}
static uint64_t levelPoints(BfhipHelm2Layout const *L, uint32_t id, uint32_t d) {
  uint32_t cnt;
  uint32_t const *ns = levelNodes(L, id, d, &cnt);
  uint64_t tot = 0;
  for (uint32_t i = 0; i < cnt; ++i) tot += L->nodes[ns[i]].i1 - L->nodes[ns[i]].i0;
  return tot;
}

/* bfHelm2RankEstForTwoCircles (src/helm2.c:14-36), C = 1, eps = 1e-15; operand order as helm2_structure.rank_est */
static int64_t rankEst(double k, double cx1, double cy1, double r1, double cx2, double cy2, double r2) {
  double const R = hypot(cx2 - cx1, cy2 - cy1);
  double const d = R - r1 - r2;
  double const p = k * r1 * r2 / d - 1 * log10(1e-15);
  return (int64_t)ceil(p);
}

static int separated(QN const *a, QN const *b) {        /* src/quadtree_node.c:393-401 */
  return hypot(a->cx - b->cx, a->cy - b->cy) > a->r + b->r + 1e1 * EPS_MACH;
}

/* allRankEstimatesAreOK, src/fac_helm2.c:510-530 */
static int ranksOk(BfhipHelm2Layout const *L, QN const *tgt, uint32_t src, uint32_t d) {
  uint32_t cnt;
  uint32_t const *ns = levelNodes(L, src, d, &cnt);
  for (uint32_t i = 0; i < cnt; ++i) {
    QN const *s = &L->nodes[ns[i]];
    if (rankEst(L->k, tgt->cx, tgt->cy, tgt->r, s->cx, s->cy, s->r) > (int64_t)(s->i1 - s->i0)) return 0;
  }
  return 1;
}

/* bfFacHelm2Prepare (src/fac_helm2.c:551-651): number of factors (0: keep dense) and the level
 * below `src` at which the source traversal starts */
static int prepare(BfhipHelm2Layout *L, uint32_t src, uint32_t tgt, uint32_t *numFactors, uint32_t *level) {
  int rc;
  if ((rc = ensureLevels(L, src)) || (rc = ensureLevels(L, tgt))) return rc;
  QN const *S = &L->nodes[src], *T = &L->nodes[tgt];
  uint32_t maxDepthBelow = 0;
  for (uint32_t d = 1; d < T->nlv && levelInternal(L, tgt, d); ++d) ++maxDepthBelow;    /* :583-591 */
  uint32_t cur = S->nlv - 1;
  while (cur > maxDepthBelow) --cur;                                                       /* :612-615 */
  while (levelPoints(L, src, cur) != S->i1 - S->i0) --cur;                                 /* :618-621 */
  while (!levelInternal(L, src, cur)) --cur;                                               /* :625-628 */
  while (cur > 0 && !ranksOk(L, T, src, cur)) --cur;                                       /* :634-638 */
  *numFactors = ranksOk(L, T, src, cur) ? cur + 2 : 0;                                     /* :642-643 */
  *level = cur;
  return 0;
}

/* bfFacHelm2Make (src/fac_helm2.c:653-704): factors built first-applied first, stored reversed */
static int makeProduct(BfhipHelm2Layout *L, uint32_t src, uint32_t tgt, uint32_t nf, uint32_t lev, uint64_t *out) {
  QN const *S = &L->nodes[src], *T = &L->nodes[tgt];
  double const k = L->k;
  int rc = 0;
  Child *ch = NULL, *factors = malloc((size_t)nf * sizeof *factors);
  int64_t *prevH = NULL, *rowH = NULL, *rank = NULL;
  uint64_t *rowOff = NULL, *colOff = NULL;
  uint32_t *sChild = NULL, *sPar = NULL, *tChild = NULL, *tPar = NULL;
  uint64_t numF = 0, numPrev = 0;
  if (!factors) { rc = bfhipFail(BFABI_ERROR_MEMORY_ERROR, "host OOM (layout)"); goto done; }

  /* makeFirstFactor (:42-160): one block per source node at level `lev` */
  {
    uint32_t cnt;
    uint32_t const *ns = levelNodes(L, src, lev, &cnt);
    ch = malloc((size_t)cnt * sizeof *ch);
    prevH = malloc((size_t)cnt * 8);
    if (!ch || !prevH) { rc = bfhipFail(BFABI_ERROR_MEMORY_ERROR, "host OOM (layout)"); goto done; }
    uint64_t r0 = 0, c0 = 0;
    for (uint32_t i = 0; i < cnt; ++i) {
      QN const *s = &L->nodes[ns[i]];
      int64_t const p = rankEst(k, s->cx, s->cy, s->r, T->cx, T->cy, T->r);
      uint64_t const npts = s->i1 - s->i0;
      uint64_t const leaf = descAdd(L, BFHIP_NODE_DENSE, (uint64_t)p, npts, NULL, 0, 0);
      addRecipe(L, leaf, BFHIP_LEAF_REEXP, nodePts(s), circPts(s, (uint64_t)p), circPts(T, (uint64_t)p));
      ch[i].node = leaf; ch[i].r0 = r0; ch[i].c0 = c0;
      r0 += (uint64_t)p; c0 += npts;
      prevH[i] = p;
    }
    factors[numF].node = descAdd(L, BFHIP_NODE_BLOCK, r0, c0, ch, cnt, BFABI_TYPE_MAT_BLOCK_DIAG);
    factors[numF].r0 = factors[numF].c0 = 0;
    ++numF;
    numPrev = cnt;
    free(ch); ch = NULL;
  }

  /* makeFactor (:222-401), i = 1 .. nf-2 */
  for (uint32_t i = 1; i + 1 < nf && !L->oom; ++i) {
    uint32_t nSrcNodes, nTgtNodes;
    uint32_t const *sp = levelNodes(L, src, lev - i, &nSrcNodes);       /* source parents */
    uint32_t const *tp = levelNodes(L, tgt, i - 1, &nTgtNodes);         /* target parents */
    uint64_t totS = 0, totT = 0;
    for (uint32_t a = 0; a < nSrcNodes; ++a) totS += L->nodes[sp[a]].nch;
    for (uint32_t a = 0; a < nTgtNodes; ++a) totT += L->nodes[tp[a]].nch;
    if (totS * nTgtNodes != numPrev) { rc = bfhipFail(BFABI_ERROR_RUNTIME_ERROR, "internal: factor shapes do not chain"); goto done; }   /* :252 */
    free(sChild); free(sPar); free(tChild); free(tPar); free(rank); free(rowH); free(rowOff); free(colOff);
    sChild = malloc((size_t)(totS + 1) * 4); sPar = malloc((size_t)(totS + 1) * 4);
    tChild = malloc((size_t)(totT + 1) * 4); tPar = malloc((size_t)(totT + 1) * 4);
    rank = malloc((size_t)(totT * totS + 1) * 8);
    rowH = calloc((size_t)(totT * nSrcNodes + 1), 8);
    rowOff = malloc((size_t)(totT * nSrcNodes + 2) * 8);
    colOff = malloc((size_t)(numPrev + 2) * 8);
    ch = malloc((size_t)(totT * totS + 1) * sizeof *ch);
    if (!sChild || !sPar || !tChild || !tPar || !rank || !rowH || !rowOff || !colOff || !ch) { rc = bfhipFail(BFABI_ERROR_MEMORY_ERROR, "host OOM (layout)"); goto done; }
    uint64_t q = 0;
    for (uint32_t a = 0; a < nSrcNodes; ++a)
      for (uint32_t c = 0; c < L->nodes[sp[a]].nch; ++c) { sChild[q] = L->nodes[sp[a]].ch[c]; sPar[q] = a; ++q; }
    q = 0;
    for (uint32_t a = 0; a < nTgtNodes; ++a)
      for (uint32_t c = 0; c < L->nodes[tp[a]].nch; ++c) { tChild[q] = L->nodes[tp[a]].ch[c]; tPar[q] = a; ++q; }
    /* rank = max(rankOr (src child circle, tgt parent circle), rankEq (src parent, tgt child)) (:291-296);
     * row height of block row (tgt child t, src parent s) = max over s's children (:305-307) */
    for (uint64_t t = 0; t < totT; ++t) {
      QN const *tc = &L->nodes[tChild[t]], *tpar = &L->nodes[tp[tPar[t]]];
      for (uint64_t s = 0; s < totS; ++s) {
        QN const *sc = &L->nodes[sChild[s]], *spar = &L->nodes[sp[sPar[s]]];
        int64_t const ror = rankEst(k, sc->cx, sc->cy, sc->r, tpar->cx, tpar->cy, tpar->r);
        int64_t const req = rankEst(k, spar->cx, spar->cy, spar->r, tc->cx, tc->cy, tc->r);
        int64_t const rk = ror > req ? ror : req;
        rank[t * totS + s] = rk;
        int64_t *hh = &rowH[t * nSrcNodes + sPar[s]];
        if (rk > *hh) *hh = rk;
      }
    }
    uint64_t const numRowsB = totT * nSrcNodes;
    rowOff[0] = 0;
    for (uint64_t b = 0; b < numRowsB; ++b) rowOff[b + 1] = rowOff[b] + (uint64_t)rowH[b];
    colOff[0] = 0;
    for (uint64_t b = 0; b < numPrev; ++b) colOff[b + 1] = colOff[b] + (uint64_t)prevH[b];
    uint64_t nc = 0;
    for (uint64_t t = 0; t < totT; ++t) {
      QN const *tc = &L->nodes[tChild[t]];
      for (uint64_t s = 0; s < totS; ++s) {
        QN const *sc = &L->nodes[sChild[s]], *spar = &L->nodes[sp[sPar[s]]];
        uint64_t const bi = t * nSrcNodes + sPar[s], bj = (uint64_t)tPar[t] * totS + s;
        uint64_t const m = (uint64_t)rowH[bi], n = (uint64_t)prevH[bj];
        uint64_t const leaf = descAdd(L, BFHIP_NODE_DENSE, m, n, NULL, 0, 0);
        /* :338-358: orig = src child circle (n points), equiv = src parent circle (m), targets = tgt child circle (m) */
        addRecipe(L, leaf, BFHIP_LEAF_REEXP, circPts(sc, n), circPts(spar, m), circPts(tc, m));
        ch[nc].node = leaf; ch[nc].r0 = rowOff[bi]; ch[nc].c0 = colOff[bj];
        ++nc;
      }
    }
    factors[numF].node = descAdd(L, BFHIP_NODE_BLOCK, rowOff[numRowsB], colOff[numPrev], ch, nc, BFABI_TYPE_MAT_BLOCK_COO);
    factors[numF].r0 = factors[numF].c0 = 0;
    ++numF;
    free(ch); ch = NULL;
    free(prevH);
    prevH = rowH; rowH = NULL;
    numPrev = numRowsB;
  }

  /* makeLastFactor (:403-509): one block per target node at level nf-2 */
  {
    uint32_t cnt;
    uint32_t const *ns = levelNodes(L, tgt, nf - 2, &cnt);
    if (cnt != numPrev) { rc = bfhipFail(BFABI_ERROR_RUNTIME_ERROR, "internal: last factor does not chain"); goto done; }              /* :419 */
    ch = malloc((size_t)(cnt + 1) * sizeof *ch);
    if (!ch) { rc = bfhipFail(BFABI_ERROR_MEMORY_ERROR, "host OOM (layout)"); goto done; }
    uint64_t r0 = 0, c0 = 0;
    for (uint32_t i = 0; i < cnt; ++i) {
      QN const *t = &L->nodes[ns[i]];
      uint64_t const n = (uint64_t)prevH[i], npts = t->i1 - t->i0;
      ch[i].node = addKernelLeaf(L, npts, n, circPts(S, n), nodePts(t));
      ch[i].r0 = r0; ch[i].c0 = c0;
      r0 += npts; c0 += n;
    }
    factors[numF].node = descAdd(L, BFHIP_NODE_BLOCK, r0, c0, ch, cnt, BFABI_TYPE_MAT_BLOCK_DIAG);
    factors[numF].r0 = factors[numF].c0 = 0;
    ++numF;
  }
  /* product order = reversed build order (:692-695) */
  for (uint64_t a = 0, b = numF - 1; a < b; ++a, --b) { Child const t = factors[a]; factors[a] = factors[b]; factors[b] = t; }
  *out = descAdd(L, BFHIP_NODE_PRODUCT, T->i1 - T->i0, S->i1 - S->i0, factors, numF, 0);
done:
  free(ch); free(factors); free(prevH); free(rowH); free(rank); free(rowOff); free(colOff);
  free(sChild); free(sPar); free(tChild); free(tPar);
  if (!rc && L->oom) rc = bfhipFail(BFABI_ERROR_MEMORY_ERROR, "host OOM (layout)");
  return rc;
}

/* facHelm2MakeMultilevel_rec (src/fac_helm2.c:860-941): a dense grid of blocks */
static int multilevelRec(BfhipHelm2Layout *L, uint32_t const *srcNodes, uint32_t nS, uint32_t const *tgtNodes, uint32_t nT,
                         Child **outCh, uint64_t *outCount, uint64_t *outRows, uint64_t *outCols) {
  Child *ch = malloc(((size_t)nS * nT + 1) * sizeof *ch);
  if (!ch) return bfhipFail(BFABI_ERROR_MEMORY_ERROR, "host OOM (layout)");
  int rc = 0;
  uint64_t nc = 0, r0 = 0, cols = 0;
  for (uint32_t ti = 0; ti < nT && !rc; ++ti) {
    uint64_t c0 = 0;
    for (uint32_t si = 0; si < nS && !rc; ++si) {
      QN const tn = L->nodes[tgtNodes[ti]], sn = L->nodes[srcNodes[si]];       /* copies: makeProduct never moves nodes, but be safe */
      uint64_t const m = tn.i1 - tn.i0, n = sn.i1 - sn.i0;
      uint64_t node = 0;
      if (m * n < MAX_DENSE_MATRIX_SIZE) {                                      /* :888 */
        node = addKernelLeaf(L, m, n, nodePts(&sn), nodePts(&tn));
      } else if (separated(&sn, &tn)) {                                         /* :890 -> :777-804 */
        uint32_t nf, lev;
        if ((rc = prepare(L, srcNodes[si], tgtNodes[ti], &nf, &lev))) break;
        if (nf == 0) node = addKernelLeaf(L, m, n, nodePts(&sn), nodePts(&tn));
        else rc = makeProduct(L, srcNodes[si], tgtNodes[ti], nf, lev, &node);
      } else {                                                                  /* :814-858 */
        Child *sub = NULL;
        uint64_t subCount = 0, rr = 0, cc = 0;
        rc = multilevelRec(L, sn.ch, sn.nch, tn.ch, tn.nch, &sub, &subCount, &rr, &cc);
        if (!rc) node = descAdd(L, BFHIP_NODE_BLOCK, rr, cc, sub, subCount, BFABI_TYPE_MAT_BLOCK_DENSE);
        free(sub);
      }
      ch[nc].node = node; ch[nc].r0 = r0; ch[nc].c0 = c0;
      ++nc;
      c0 += n;
    }
    r0 += L->nodes[tgtNodes[ti]].i1 - L->nodes[tgtNodes[ti]].i0;
  }
  for (uint32_t si = 0; si < nS; ++si) cols += L->nodes[srcNodes[si]].i1 - L->nodes[srcNodes[si]].i0;
  if (!rc && L->oom) rc = bfhipFail(BFABI_ERROR_MEMORY_ERROR, "host OOM (layout)");
  if (rc) { free(ch); return rc; }
  *outCh = ch; *outCount = nc; *outRows = r0; *outCols = cols;
  return 0;
}

void bfhipHelm2LayoutFree(BfhipHelm2Layout **pl) {
  if (!pl || !*pl) return;
  BfhipHelm2Layout *L = *pl;
  for (uint32_t i = 0; i < L->numNodes; ++i) { free(L->nodes[i].lv); free(L->nodes[i].lvBegin); }
  free(L->nodes); free(L->perm); free(L->treePoints); free(L->tperm); free(L->ttreePoints);
  free(L->kind); free(L->blockKind); free(L->rows); free(L->cols); free(L->childBegin);
  free(L->childNode); free(L->childRow0); free(L->childCol0); free(L->topRowBlock); free(L->recipes);
  free(L);
  *pl = NULL;
}

int bfhipHelm2LayoutCreate2(double const *points, uint64_t numPoints, double const *tgtPoints, uint64_t numTgtPoints, double wavenumber,
                            BfhipHelm2Layout **out) {
  if (!points || !out) return bfhipFail(BFABI_ERROR_INVALID_ARGUMENTS, "NULL argument");
  *out = NULL;
  if (numPoints < 2 || numPoints > 0x7fffffffu || (tgtPoints && (numTgtPoints < 2 || numTgtPoints > 0x7fffffffu)))
    return bfhipFail(BFABI_ERROR_INVALID_ARGUMENTS, "numPoints out of range");
  if (!(wavenumber > 0)) return bfhipFail(BFABI_ERROR_INVALID_ARGUMENTS, "wavenumber must be positive");
  BfhipHelm2Layout *L = calloc(1, sizeof *L);
  if (!L) return bfhipFail(BFABI_ERROR_MEMORY_ERROR, "host OOM");
  L->n = numPoints; L->k = wavenumber;
  L->perm = malloc((size_t)numPoints * 8);
  L->treePoints = malloc((size_t)numPoints * 16);
  int rc = (!L->perm || !L->treePoints) ? bfhipFail(BFABI_ERROR_MEMORY_ERROR, "host OOM") : 0;
  if (!rc && tgtPoints) {
    L->m = numTgtPoints;
    L->tperm = malloc((size_t)numTgtPoints * 8);
    L->ttreePoints = malloc((size_t)numTgtPoints * 16);
    if (!L->tperm || !L->ttreePoints) rc = bfhipFail(BFABI_ERROR_MEMORY_ERROR, "host OOM");
  }
  Child *ch = NULL;
  uint32_t sroot = 0, troot = 0;
  if (!rc) rc = buildQuadtree(L, points, numPoints, L->perm, 0, &sroot);
  if (!rc) {
    for (uint64_t i = 0; i < numPoints; ++i) { L->treePoints[2 * i] = points[2 * L->perm[i]]; L->treePoints[2 * i + 1] = points[2 * L->perm[i] + 1]; }
    troot = sroot;
    if (tgtPoints) {
      rc = buildQuadtree(L, tgtPoints, numTgtPoints, L->tperm, 1, &troot);
      if (!rc)
        for (uint64_t i = 0; i < numTgtPoints; ++i) { L->ttreePoints[2 * i] = tgtPoints[2 * L->tperm[i]]; L->ttreePoints[2 * i + 1] = tgtPoints[2 * L->tperm[i] + 1]; }
    }
  }
  if (!rc) rc = ensureLevels(L, sroot);
  if (!rc) rc = ensureLevels(L, troot);
  if (!rc && (L->nodes[sroot].nlv < 3 || L->nodes[troot].nlv < 3)) rc = bfhipFail(BFABI_ERROR_INVALID_ARGUMENTS, "quadtree has fewer than 3 levels");
  if (!rc) {
    /* the top level of the operator is the grid of level-2 nodes (:956-982): rows = target nodes */
    uint32_t cntS, cntT;
    uint32_t const *s2 = levelNodes(L, sroot, 2, &cntS), *t2 = levelNodes(L, troot, 2, &cntT);
    uint32_t *src2 = malloc((size_t)cntS * 4), *tgt2 = malloc((size_t)cntT * 4);
    uint64_t nc = 0, rr = 0, cc = 0;
    if (!src2 || !tgt2) rc = bfhipFail(BFABI_ERROR_MEMORY_ERROR, "host OOM");
    if (!rc) {
      memcpy(src2, s2, (size_t)cntS * 4);
      memcpy(tgt2, t2, (size_t)cntT * 4);
      rc = multilevelRec(L, src2, cntS, tgt2, cntT, &ch, &nc, &rr, &cc);
    }
    if (!rc) {
      uint64_t const root = descAdd(L, BFHIP_NODE_BLOCK, rr, cc, ch, nc, BFABI_TYPE_MAT_BLOCK_DENSE);
      L->topRowBlock = malloc((size_t)(nc + 1) * 8);
      if (!L->topRowBlock || L->oom) rc = bfhipFail(BFABI_ERROR_MEMORY_ERROR, "host OOM");
      else {
        for (uint64_t i = 0; i < nc; ++i) L->topRowBlock[i] = i / cntS;
        BfhipDesc *d = &L->desc;
        memset(d, 0, sizeof *d);
        d->structSize = sizeof *d; d->dtype = BFHIP_C128; d->numNodes = L->numDesc; d->root = root;
        d->kind = L->kind; d->rows = L->rows; d->cols = L->cols; d->childBegin = L->childBegin;
        d->childNode = L->childNode; d->childRow0 = L->childRow0; d->childCol0 = L->childCol0;
        d->topRowBlock = L->topRowBlock; d->blockKind = L->blockKind;
      }
    }
    free(src2); free(tgt2);
  }
  free(ch);
  if (rc) { bfhipHelm2LayoutFree(&L); return rc; }
  *out = L;
  return 0;
}

/* bfFacHelm2MakeSingleLevel (src/fac_helm2.c:706-729; examples/simple/bf_one_block.c:162): the butterfly
 * of ONE (source node, target node) pair of the quadtree on `points`.  The nodes are named by their
 * paths from the root: child positions among the NON-EMPTY children, in quadrant order.  The operator
 * maps the source node's points to the target node's points (tree order); the recipes address the whole
 * tree-ordered point array. */
int bfhipHelm2LayoutCreateSingle(double const *points, uint64_t numPoints, double wavenumber, uint32_t const *srcPath, uint32_t srcDepth,
                                 uint32_t const *tgtPath, uint32_t tgtDepth, BfhipHelm2Layout **out) {
  if (!points || !out || (srcDepth && !srcPath) || (tgtDepth && !tgtPath)) return bfhipFail(BFABI_ERROR_INVALID_ARGUMENTS, "NULL argument");
  *out = NULL;
  if (numPoints < 2 || numPoints > 0x7fffffffu) return bfhipFail(BFABI_ERROR_INVALID_ARGUMENTS, "numPoints out of range");
  if (!(wavenumber > 0)) return bfhipFail(BFABI_ERROR_INVALID_ARGUMENTS, "wavenumber must be positive");
  BfhipHelm2Layout *L = calloc(1, sizeof *L);
  if (!L) return bfhipFail(BFABI_ERROR_MEMORY_ERROR, "host OOM");
  L->n = numPoints; L->k = wavenumber;
  L->perm = malloc((size_t)numPoints * 8);
  L->treePoints = malloc((size_t)numPoints * 16);
  int rc = (!L->perm || !L->treePoints) ? bfhipFail(BFABI_ERROR_MEMORY_ERROR, "host OOM") : 0;
  uint32_t root = 0;
  if (!rc) rc = buildQuadtree(L, points, numPoints, L->perm, 0, &root);
  uint32_t sn = root, tn = root;
  for (uint32_t d = 0; d < srcDepth && !rc; ++d) {
    if (srcPath[d] >= L->nodes[sn].nch) rc = bfhipFail(BFABI_ERROR_INVALID_ARGUMENTS, "source path leaves the tree at depth %u", d);
    else sn = L->nodes[sn].ch[srcPath[d]];
  }
  for (uint32_t d = 0; d < tgtDepth && !rc; ++d) {
    if (tgtPath[d] >= L->nodes[tn].nch) rc = bfhipFail(BFABI_ERROR_INVALID_ARGUMENTS, "target path leaves the tree at depth %u", d);
    else tn = L->nodes[tn].ch[tgtPath[d]];
  }
  if (!rc && srcDepth != tgtDepth) rc = bfhipFail(BFABI_ERROR_INVALID_ARGUMENTS, "source and target nodes must be on the same level");
  if (!rc) {
    for (uint64_t i = 0; i < numPoints; ++i) { L->treePoints[2 * i] = points[2 * L->perm[i]]; L->treePoints[2 * i + 1] = points[2 * L->perm[i] + 1]; }
    uint32_t nf = 0, lev = 0;
    if (L->nodes[sn].nch == 0 || L->nodes[tn].nch == 0) rc = bfhipFail(BFABI_ERROR_INVALID_ARGUMENTS, "a leaf node has no butterfly");
    if (!rc) rc = prepare(L, sn, tn, &nf, &lev);
    if (!rc && nf == 0) rc = bfhipFail(BFABI_ERROR_INVALID_ARGUMENTS, "this node pair is not butterfliable (rank estimates exceed the point counts)");
    uint64_t prod = 0;
    if (!rc) rc = makeProduct(L, sn, tn, nf, lev, &prod);
    if (!rc) {
      BfhipDesc *d = &L->desc;
      memset(d, 0, sizeof *d);
      d->structSize = sizeof *d; d->dtype = BFHIP_C128; d->numNodes = L->numDesc; d->root = prod;
      d->kind = L->kind; d->rows = L->rows; d->cols = L->cols; d->childBegin = L->childBegin;
      d->childNode = L->childNode; d->childRow0 = L->childRow0; d->childCol0 = L->childCol0; d->blockKind = L->blockKind;
    }
  }
  if (rc) { bfhipHelm2LayoutFree(&L); return rc; }
  *out = L;
  return 0;
}

int bfhipHelm2LayoutCreate(double const *points, uint64_t numPoints, double wavenumber, BfhipHelm2Layout **out) {
  return bfhipHelm2LayoutCreate2(points, numPoints, NULL, 0, wavenumber, out);
}

BfhipDesc const *bfhipHelm2LayoutGetDesc(BfhipHelm2Layout const *L) { return L ? &L->desc : NULL; }
BfhipHelm2Recipe const *bfhipHelm2LayoutGetRecipes(BfhipHelm2Layout const *L, uint64_t *count) {
  if (count) *count = L ? L->numRecipes : 0;
  return L ? L->recipes : NULL;
}
uint64_t const *bfhipHelm2LayoutGetPerm(BfhipHelm2Layout const *L) { return L ? L->perm : NULL; }
double const *bfhipHelm2LayoutGetTreePoints(BfhipHelm2Layout const *L) { return L ? L->treePoints : NULL; }
uint64_t const *bfhipHelm2LayoutGetTgtPerm(BfhipHelm2Layout const *L) { return L ? L->tperm : NULL; }
double const *bfhipHelm2LayoutGetTgtTreePoints(BfhipHelm2Layout const *L) { return L ? L->ttreePoints : NULL; }

/* points -> device operator in one call: what bfFacHelm2MakeMultilevel (src/fac_helm2.c:943-1002)
 * followed by bfhipCompile gives, without the CPU build */
int bfhipFacHelm2MakeMultilevel2(double const *points, double const *normals, double const *colWeights, uint64_t numPoints,
                                 double const *tgtPoints, double const *tgtNormals, uint64_t numTgtPoints,
                                 BfhipHelm2Problem const *params, BfhipOptions const *opts, BfhipOperator **out, uint64_t *permOut,
                                 uint64_t *tgtPermOut, BfhipBuildStats *stats) {
  if (!points || !params || !out) return bfhipFail(BFABI_ERROR_INVALID_ARGUMENTS, "NULL argument");
  *out = NULL;
  if (params->structSize < sizeof *params) return bfhipFail(BFABI_ERROR_INVALID_ARGUMENTS, "BfhipHelm2Problem.structSize too small");
  BfhipHelm2Layout *L = NULL;
  int rc = bfhipHelm2LayoutCreate2(points, numPoints, tgtPoints, tgtPoints ? numTgtPoints : 0, params->wavenumber, &L);
  if (rc) return rc;
  double *nrm = NULL, *w = NULL, *tnrm = NULL;
  if (normals) {
    nrm = malloc((size_t)numPoints * 16);
    if (!nrm) rc = bfhipFail(BFABI_ERROR_MEMORY_ERROR, "host OOM");
    else for (uint64_t i = 0; i < numPoints; ++i) { nrm[2 * i] = normals[2 * L->perm[i]]; nrm[2 * i + 1] = normals[2 * L->perm[i] + 1]; }
  }
  if (!rc && colWeights) {
    w = malloc((size_t)numPoints * 8);
    if (!w) rc = bfhipFail(BFABI_ERROR_MEMORY_ERROR, "host OOM");
    else for (uint64_t i = 0; i < numPoints; ++i) w[i] = colWeights[L->perm[i]];
  }
  if (!rc && tgtPoints && tgtNormals) {
    tnrm = malloc((size_t)numTgtPoints * 16);
    if (!tnrm) rc = bfhipFail(BFABI_ERROR_MEMORY_ERROR, "host OOM");
    else for (uint64_t i = 0; i < numTgtPoints; ++i) { tnrm[2 * i] = tgtNormals[2 * L->tperm[i]]; tnrm[2 * i + 1] = tgtNormals[2 * L->tperm[i] + 1]; }
  }
  if (!rc) {
    BfhipHelm2Problem prob = *params;
    prob.structSize = sizeof prob;
    prob.points = L->treePoints; prob.numPoints = numPoints; prob.recipes = L->recipes; prob.numRecipes = L->numRecipes;
    prob.normals = nrm; prob.colWeights = w; prob.origIndex = L->perm;
    prob.tgtPoints = tgtPoints ? L->ttreePoints : NULL; prob.numTgtPoints = tgtPoints ? numTgtPoints : 0; prob.tgtNormals = tnrm;
    rc = bfhipBuildHelm2(&L->desc, &prob, opts, out, stats);
  }
  if (!rc && permOut) memcpy(permOut, L->perm, (size_t)numPoints * 8);
  if (!rc && tgtPermOut && tgtPoints) memcpy(tgtPermOut, L->tperm, (size_t)numTgtPoints * 8);
  free(nrm); free(w); free(tnrm);
  bfhipHelm2LayoutFree(&L);
  return rc;
}

int bfhipFacHelm2MakeMultilevel(double const *points, double const *normals, double const *colWeights, uint64_t numPoints,
                                BfhipHelm2Problem const *params, BfhipOptions const *opts, BfhipOperator **out, uint64_t *permOut,
                                BfhipBuildStats *stats) {
  return bfhipFacHelm2MakeMultilevel2(points, normals, colWeights, numPoints, NULL, NULL, 0, params, opts, out, permOut, NULL, stats);
}
