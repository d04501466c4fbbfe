// bfhip_shard.hip -- the multi-GPU step behind the C-ABI: one process per GPU,
// each holding the operator of its share of the top-level blocks, and ONE RCCL collective per apply.
//
// The reference computes every top-level block row independently from the full x
// (bfMatBlockDenseMul, src/mat_block_dense.c:534-566), so
//   "rows"   : a rank owns whole block rows; its operator writes their rows, compacted, straight into
//              its slot of a rank-major gather buffer; ONE in-place ncclAllGather over xGMI; one
//              tiny kernel puts the <= 16 row segments into global row order in the caller's y.
//   "blocks" : a rank owns (row, col) blocks at their original offsets (finer balance: 144 blocks
//              instead of 12 rows on a circle); its operator writes a full-length partial y; ONE
//              in-place ncclAllReduce (sum).
// Both are enqueued on the caller's stream right behind the stage kernels: no host synchronisation.
//
// RCCL is resolved at run time (dlopen of the copy already in the process -- PyTorch ships one --
// else the system one): plain C hosts without RCCL still load libbfhip.so; the sharded entry points
// then fail with RUNTIME_ERROR instead of the library failing to load.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <dlfcn.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "bfhip_internal.h"
#include "../../include/bfhip_abi.h"

namespace {

struct Rccl {
  void *lib;
  ncclResult_t (*GetUniqueId)(ncclUniqueId *);
  ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int);
  ncclResult_t (*CommDestroy)(ncclComm_t);
  ncclResult_t (*CommAbort)(ncclComm_t);
  ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t);
  ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t);
  const char *(*GetErrorString)(ncclResult_t);
};
Rccl g;

int loadRccl() {
  if (g.lib) return 0;
  char const *names[] = {"librccl.so.1", "librccl.so"};
  void *lib = NULL;
  for (int pass = 0; pass < 2 && !lib; ++pass)         // pass 0: a copy already mapped (torch's); pass 1: load one
    for (int i = 0; i < 2 && !lib; ++i) lib = dlopen(names[i], RTLD_NOW | RTLD_GLOBAL | (pass == 0 ? RTLD_NOLOAD : 0));
  if (!lib) return bfhipFail(BFABI_ERROR_RUNTIME_ERROR, "RCCL not found (librccl.so.1): %s", dlerror());
#define SYM(field, name) do { *(void **)&g.field = dlsym(lib, name); if (!g.field) return bfhipFail(BFABI_ERROR_RUNTIME_ERROR, "RCCL lacks %s", name); } while (0)
  SYM(GetUniqueId, "ncclGetUniqueId"); SYM(CommInitRank, "ncclCommInitRank"); SYM(CommDestroy, "ncclCommDestroy"); SYM(CommAbort, "ncclCommAbort");
  SYM(AllGather, "ncclAllGather"); SYM(AllReduce, "ncclAllReduce"); SYM(GetErrorString, "ncclGetErrorString");
#undef SYM
  g.lib = lib;
  return 0;
}

int ncclFail(ncclResult_t r, char const *what) {
  if (r == ncclSuccess) return 0;
  return bfhipFail(BFABI_ERROR_RUNTIME_ERROR, "%s: %s", what, g.GetErrorString ? g.GetErrorString(r) : "RCCL error");
}
int hipFailS(hipError_t e, char const *what) {
  if (e == hipSuccess) return 0;
  return bfhipFail(e == hipErrorOutOfMemory ? BFABI_ERROR_MEMORY_ERROR : BFABI_ERROR_RUNTIME_ERROR, "%s: %s", what, hipGetErrorString(e));
}

struct Seg { uint64_t globalOff, srcOff, rows; };    // rows [globalOff, +rows) of y come from gather-buffer rows [srcOff, +rows)

// Row ranges that several ranks contributed to (a block row shared by columns): y[row] = sum, in list order, of the
// partials.  groups are sorted by globalOff and disjoint; group g's sources are srcOff[srcBegin[g] .. srcBegin[g + 1]).
// One thread per scalar (double / float) of y; a range with one source is a plain copy.
struct SumGroup { uint64_t globalOff, rows; uint32_t srcBegin, srcEnd; };
template <typename T>
__global__ __launch_bounds__(256) void bfSumSegmentsKernel(SumGroup const *groups, uint32_t numGroups, uint64_t const *srcOff, uint64_t numRows,
                                                           T const *gathered, T *y, uint64_t scalarsPerRow) {
  uint64_t const u = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (u >= numRows * scalarsPerRow) return;
  uint64_t const row = u / scalarsPerRow, k = u - row * scalarsPerRow;
  uint32_t lo = 0, hi = numGroups;
  while (hi - lo > 1) { uint32_t mid = (lo + hi) / 2; if (groups[mid].globalOff <= row) lo = mid; else hi = mid; }
  SumGroup const g = groups[lo];
  uint64_t const r = row - g.globalOff;
  T acc = gathered[(srcOff[g.srcBegin] + r) * scalarsPerRow + k];
  for (uint32_t s = g.srcBegin + 1; s < g.srcEnd; ++s) acc += gathered[(srcOff[s] + r) * scalarsPerRow + k];      // fixed order
  y[u] = acc;
}

// y[row][:] = gathered[srcOff(seg) + row - globalOff(seg)][:], one thread per unit of the row
// (unit = 16 bytes when the row size allows, else 8 or 4); segments are sorted by globalOff
template <typename U>
__global__ __launch_bounds__(256) void bfScatterSegmentsKernel(Seg const *segs, uint32_t numSegs, uint64_t numRows,
                                                               U const *gathered, U *y, uint64_t unitsPerRow) {
  uint64_t u = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (u >= numRows * unitsPerRow) return;
  uint64_t const row = u / unitsPerRow, k = u - row * unitsPerRow;
  uint32_t lo = 0, hi = numSegs;                       // last segment with globalOff <= row
  while (hi - lo > 1) { uint32_t mid = (lo + hi) / 2; if (segs[mid].globalOff <= row) lo = mid; else hi = mid; }
  y[u] = gathered[(segs[lo].srcOff + (row - segs[lo].globalOff)) * unitsPerRow + k];
}

// the inverse of a rank's compaction, for the adjoint step: vr[local row][:] = v[globalOff(seg) + local row - srcOff(seg)][:]
// over THIS rank's segments (srcOff = first local row of the segment; sorted by it)
template <typename U>
__global__ __launch_bounds__(256) void bfGatherSegmentsKernel(Seg const *segs, uint32_t numSegs, uint64_t numLocalRows,
                                                              U const *v, U *vr, uint64_t unitsPerRow) {
  uint64_t u = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (u >= numLocalRows * unitsPerRow) return;
  uint64_t const row = u / unitsPerRow, k = u - row * unitsPerRow;
  uint32_t lo = 0, hi = numSegs;                       // last segment with srcOff <= row
  while (hi - lo > 1) { uint32_t mid = (lo + hi) / 2; if (segs[mid].srcOff <= row) lo = mid; else hi = mid; }
  vr[u] = v[(segs[lo].globalOff + (row - segs[lo].srcOff)) * unitsPerRow + k];
}

}  // namespace

struct BfhipComm { ncclComm_t comm; int nranks, rank, device; int aborted; };

struct BfhipSharded {
  BfhipOperator *op;
  BfhipComm *comm;
  int device;                    // copied at create time: freeing never dereferences the communicator
  uint32_t mode, dtype, elemSize, maxRhs, numSegs;
  uint64_t numRowsGlobal, myRows, maxRows;
  void *dGather;                 // rows mode: nranks * maxRows * maxRhs elements
  Seg *dSegs;
  Seg *hSegs;
  SumGroup *dGroups;             // rows mode with shared ranges: the partials of a range are added after the gather
  uint64_t *dSrcOff;
  uint32_t numGroups;
  hipEvent_t e0, e1, e2;
  int timed;
  int timing;       // record the three events (default on)
  // the adjoint step (operators compiled with an adjoint plan)
  uint64_t numCols;              // columns of the operator = length of x and of A^T v
  int hasAdjoint;
  Seg *dMySegs;                  // rows mode: this rank's segments, srcOff = first LOCAL row (the order its operator produces them)
  uint32_t numMySegs;
  uint64_t myFirstGlobal;        // one segment only: its rows of v are used in place
  void *dVr;                     // rows mode, several segments: this rank's entries of v, compacted (myRows x maxRhs)
  void *dCov;                    // covariance products of real operators: two vectors of the longer side
  void *dHostX, *dHostY;         // staging of the host-vector entry (the vtable shim): the longer side x maxRhs each
};

extern "C" {

int bfhipCommGetUniqueId(void *id128) {
  if (!id128) return bfhipFail(BFABI_ERROR_INVALID_ARGUMENTS, "NULL id buffer");
  int rc = loadRccl();
  if (rc) return rc;
  static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
  return ncclFail(g.GetUniqueId((ncclUniqueId *)id128), "ncclGetUniqueId");
}

int bfhipCommInitRank(void const *id128, int nranks, int rank, int device, BfhipComm **out) {
  if (!id128 || !out || nranks < 1 || rank < 0 || rank >= nranks) return bfhipFail(BFABI_ERROR_INVALID_ARGUMENTS, "bad communicator arguments");
  *out = NULL;
  int rc = loadRccl();
  if (rc) return rc;
  int prev = -1;
  (void)hipGetDevice(&prev);
  if (device >= 0 && (rc = hipFailS(hipSetDevice(device), "hipSetDevice"))) return rc;
  BfhipComm *c = (BfhipComm *)calloc(1, sizeof *c);
  if (!c) return bfhipFail(BFABI_ERROR_MEMORY_ERROR, "host OOM");
  ncclUniqueId id;
  memcpy(&id, id128, sizeof id);
  rc = ncclFail(g.CommInitRank(&c->comm, nranks, id, rank), "ncclCommInitRank");
  if (!rc) rc = hipFailS(hipGetDevice(&c->device), "hipGetDevice");
  if (device >= 0 && prev >= 0 && prev != device) (void)hipSetDevice(prev);
  if (rc) { free(c); return rc; }
  c->nranks = nranks; c->rank = rank;
  *out = c;
  return 0;
}

void bfhipCommDestroy(BfhipComm **pc) {
  if (!pc || !*pc) return;
  if ((*pc)->comm && !(*pc)->aborted && g.CommDestroy) (void)g.CommDestroy((*pc)->comm);      // an aborted communicator is already gone
  free(*pc);
  *pc = NULL;
}

void bfhipShardedFree(BfhipSharded **ps) {
  if (!ps || !*ps) return;
  BfhipSharded *s = *ps;
  int prev = -1;
  (void)hipGetDevice(&prev);
  (void)hipSetDevice(s->device);
  (void)hipFree(s->dGather); (void)hipFree(s->dSegs); (void)hipFree(s->dGroups); (void)hipFree(s->dSrcOff);
  (void)hipFree(s->dMySegs); (void)hipFree(s->dVr); (void)hipFree(s->dCov); (void)hipFree(s->dHostX); (void)hipFree(s->dHostY);
  if (s->e0) (void)hipEventDestroy(s->e0);
  if (s->e1) (void)hipEventDestroy(s->e1);
  if (s->e2) (void)hipEventDestroy(s->e2);
  free(s->hSegs);
  free(s);
  *ps = NULL;
  if (prev >= 0) (void)hipSetDevice(prev);
}

int bfhipShardedCreate(BfhipOperator *op, BfhipComm *comm, BfhipShardSpec const *spec, uint32_t maxRhs, BfhipSharded **out) {
  if (!op || !comm || !spec || !out || spec->structSize < BFHIP_SHARDSPEC_SIZE_V1) return bfhipFail(BFABI_ERROR_INVALID_ARGUMENTS, "bad sharded-apply arguments");
  uint64_t const *segGlobalOff = spec->structSize >= sizeof(BfhipShardSpec) ? spec->segGlobalOff : NULL;
  *out = NULL;
  if (spec->mode != BFHIP_SHARD_ROWS && spec->mode != BFHIP_SHARD_BLOCKS) return bfhipFail(BFABI_ERROR_INVALID_ARGUMENTS, "unknown shard mode");
  BfhipStats st;
  memset(&st, 0, sizeof st);
  st.structSize = sizeof st;
  int rc = bfhipGetStats(op, &st);
  if (rc) return rc;
  if (bfhipOperatorDevice(op) != comm->device) return bfhipFail(BFABI_ERROR_INVALID_ARGUMENTS, "operator lives on device %d, communicator on %d", bfhipOperatorDevice(op), comm->device);
  BfhipSharded *s = (BfhipSharded *)calloc(1, sizeof *s);
  if (!s) return bfhipFail(BFABI_ERROR_MEMORY_ERROR, "host OOM");
  s->op = op; s->comm = comm; s->device = comm->device; s->mode = spec->mode; s->dtype = st.dtype;
  s->elemSize = st.dtype == BFHIP_C128 ? 16 : st.dtype == BFHIP_F64 ? 8 : 4;
  s->maxRhs = maxRhs ? maxRhs : 1;
  s->numRowsGlobal = spec->numRowsGlobal;
  int prev = -1;
  (void)hipGetDevice(&prev);
  if ((rc = hipFailS(hipSetDevice(comm->device), "hipSetDevice"))) { free(s); return rc; }
  if (spec->mode == BFHIP_SHARD_BLOCKS) {
    if (st.numRows != spec->numRowsGlobal) rc = bfhipFail(BFABI_ERROR_INCOMPATIBLE_SHAPES, "blocks mode: the local operator must produce all %llu rows", (unsigned long long)spec->numRowsGlobal);
  } else {
    // segments in global row order; a rank's compact local order is its segments in that same order
    uint64_t *rowsOf = (uint64_t *)calloc((size_t)comm->nranks, 8), pos = 0;
    s->hSegs = (Seg *)malloc((spec->numSegments ? spec->numSegments : 1) * sizeof(Seg));
    if (!rowsOf || !s->hSegs) rc = bfhipFail(BFABI_ERROR_MEMORY_ERROR, "host OOM");
    for (uint32_t i = 0; i < spec->numSegments && !rc; ++i) {
      if (!spec->segRows || !spec->segOwner || spec->segOwner[i] >= (uint32_t)comm->nranks) { rc = bfhipFail(BFABI_ERROR_INVALID_ARGUMENTS, "segment %u: bad owner", i); break; }
      rowsOf[spec->segOwner[i]] += spec->segRows[i];
      pos += spec->segRows[i];
    }
    if (!rc && !segGlobalOff && pos != spec->numRowsGlobal) rc = bfhipFail(BFABI_ERROR_INCOMPATIBLE_SHAPES, "segments cover %llu rows, operator has %llu", (unsigned long long)pos, (unsigned long long)spec->numRowsGlobal);
    if (!rc) {
      for (int r = 0; r < comm->nranks; ++r) if (rowsOf[r] > s->maxRows) s->maxRows = rowsOf[r];
      s->myRows = rowsOf[comm->rank];
      if (st.numRows != s->myRows) rc = bfhipFail(BFABI_ERROR_INCOMPATIBLE_SHAPES, "rows mode: the local operator produces %llu rows, this rank's segments hold %llu", (unsigned long long)st.numRows, (unsigned long long)s->myRows);
    }
    if (!rc) {
      memset(rowsOf, 0, (size_t)comm->nranks * 8);
      uint64_t g0 = 0;
      for (uint32_t i = 0; i < spec->numSegments; ++i) {
        uint32_t o = spec->segOwner[i];
        s->hSegs[i].globalOff = segGlobalOff ? segGlobalOff[i] : g0; s->hSegs[i].srcOff = (uint64_t)o * s->maxRows + rowsOf[o]; s->hSegs[i].rows = spec->segRows[i];
        rowsOf[o] += spec->segRows[i]; g0 += spec->segRows[i];
      }
      s->numSegs = spec->numSegments;
      if (segGlobalOff) {
        // ranges in global order (stable: list order inside a range); identical ranges form one group, anything else
        // that overlaps, or a row nobody computes, is refused
        uint32_t const ns = spec->numSegments;
        uint32_t *ord = (uint32_t *)malloc((ns ? ns : 1) * sizeof *ord);
        SumGroup *hg = (SumGroup *)malloc((ns ? ns : 1) * sizeof *hg);
        uint64_t *hsrc = (uint64_t *)malloc((ns ? ns : 1) * 8);
        if (!ord || !hg || !hsrc) rc = bfhipFail(BFABI_ERROR_MEMORY_ERROR, "host OOM");
        for (uint32_t i = 0; i < ns && !rc; ++i) ord[i] = i;
        for (uint32_t i = 1; i < ns && !rc; ++i) {            // insertion sort by globalOff: segment lists are short (<= 16 ranges x ranks)
          uint32_t v = ord[i], j = i;
          while (j > 0 && s->hSegs[ord[j - 1]].globalOff > s->hSegs[v].globalOff) { ord[j] = ord[j - 1]; --j; }
          ord[j] = v;
        }
        uint32_t ng = 0;
        uint64_t covered = 0;
        int shared = 0;
        for (uint32_t i = 0; i < ns && !rc; ++i) {
          Seg const *sg = &s->hSegs[ord[i]];
          if (!sg->rows) continue;
          if (ng && hg[ng - 1].globalOff == sg->globalOff && hg[ng - 1].rows == sg->rows) { hsrc[hg[ng - 1].srcEnd++] = sg->srcOff; shared = 1; continue; }
          if (sg->globalOff != covered) { rc = bfhipFail(BFABI_ERROR_INVALID_ARGUMENTS, "segment %u: ranges must tile the rows (identical ranges may repeat); row %llu", ord[i], (unsigned long long)covered); break; }
          hg[ng].globalOff = sg->globalOff; hg[ng].rows = sg->rows;
          hg[ng].srcBegin = ng ? hg[ng - 1].srcEnd : 0; hg[ng].srcEnd = hg[ng].srcBegin + 1;
          hsrc[hg[ng].srcBegin] = sg->srcOff;
          covered += sg->rows;
          ++ng;
        }
        if (!rc && covered != spec->numRowsGlobal) rc = bfhipFail(BFABI_ERROR_INCOMPATIBLE_SHAPES, "segments cover %llu rows, operator has %llu", (unsigned long long)covered, (unsigned long long)spec->numRowsGlobal);
        if (!rc && shared) {
          s->numGroups = ng;
          rc = hipFailS(hipMalloc((void **)&s->dGroups, ng * sizeof *hg), "hipMalloc(sum groups)");
          if (!rc) rc = hipFailS(hipMalloc((void **)&s->dSrcOff, (size_t)ns * 8), "hipMalloc(sum sources)");
          if (!rc) rc = hipFailS(hipMemcpy(s->dGroups, hg, ng * sizeof *hg, hipMemcpyHostToDevice), "hipMemcpy(sum groups)");
          if (!rc) rc = hipFailS(hipMemcpy(s->dSrcOff, hsrc, (size_t)ns * 8, hipMemcpyHostToDevice), "hipMemcpy(sum sources)");
        } else if (!rc) {
          // every range has one owner after all: the plain reorder kernel needs the segments sorted by global row
          Seg *sorted = (Seg *)malloc((ns ? ns : 1) * sizeof *sorted);
          if (!sorted) rc = bfhipFail(BFABI_ERROR_MEMORY_ERROR, "host OOM");
          else { for (uint32_t i = 0; i < ns; ++i) sorted[i] = s->hSegs[ord[i]]; memcpy(s->hSegs, sorted, ns * sizeof *sorted); free(sorted); }
        }
        free(ord); free(hg); free(hsrc);
      }
      if (!rc) rc = hipFailS(hipMalloc(&s->dGather, (size_t)comm->nranks * s->maxRows * s->maxRhs * s->elemSize + 16), "hipMalloc(gather buffer)");
      if (!rc) rc = hipFailS(hipMalloc((void **)&s->dSegs, (s->numSegs ? s->numSegs : 1) * sizeof(Seg)), "hipMalloc(segments)");
      if (!rc) rc = hipFailS(hipMemcpy(s->dSegs, s->hSegs, s->numSegs * sizeof(Seg), hipMemcpyHostToDevice), "hipMemcpy(segments)");
    }
    free(rowsOf);
  }
  s->numCols = st.numCols;
  s->hasAdjoint = bfhipOperatorHasAdjoint(op);
  if (!rc && s->hasAdjoint && spec->mode == BFHIP_SHARD_ROWS) {
    // this rank's segments in the order its operator yields them (list order), with their local offsets
    uint32_t cnt = 0;
    for (uint32_t i = 0; i < spec->numSegments; ++i) cnt += spec->segOwner[i] == (uint32_t)comm->rank && spec->segRows[i] > 0;
    Seg *mine = (Seg *)malloc((cnt ? cnt : 1) * sizeof(Seg));
    if (!mine) rc = bfhipFail(BFABI_ERROR_MEMORY_ERROR, "host OOM");
    else {
      uint64_t local = 0, g0 = 0;
      uint32_t k = 0;
      for (uint32_t i = 0; i < spec->numSegments; ++i) {
        uint64_t const goff = segGlobalOff ? segGlobalOff[i] : g0;
        g0 += spec->segRows[i];
        if (spec->segOwner[i] != (uint32_t)comm->rank || !spec->segRows[i]) continue;
        mine[k].globalOff = goff; mine[k].srcOff = local; mine[k].rows = spec->segRows[i];
        local += spec->segRows[i]; ++k;
      }
      s->numMySegs = cnt;
      // consecutive segments that are consecutive in v as well are one run: a rank that owns ONE run reads v in place
      int oneRun = cnt > 0;
      for (uint32_t i = 1; i < cnt; ++i) oneRun = oneRun && mine[i].globalOff == mine[i - 1].globalOff + mine[i - 1].rows;
      if (oneRun) { s->numMySegs = 1; s->myFirstGlobal = mine[0].globalOff; }
      else if (cnt) {
        rc = hipFailS(hipMalloc((void **)&s->dMySegs, cnt * sizeof(Seg)), "hipMalloc(own segments)");
        if (!rc) rc = hipFailS(hipMemcpy(s->dMySegs, mine, cnt * sizeof(Seg), hipMemcpyHostToDevice), "hipMemcpy(own segments)");
        if (!rc) rc = hipFailS(hipMalloc(&s->dVr, (size_t)s->myRows * s->maxRhs * s->elemSize + 16), "hipMalloc(compacted v)");
      }
      free(mine);
    }
  }
  {
    uint64_t const big = s->numRowsGlobal > s->numCols ? s->numRowsGlobal : s->numCols;
    if (!rc && s->hasAdjoint && s->dtype != BFHIP_C128) rc = hipFailS(hipMalloc(&s->dCov, 2 * (size_t)big * s->elemSize + 32), "hipMalloc(covariance scratch)");
    if (!rc) rc = hipFailS(hipMalloc(&s->dHostX, (size_t)big * s->maxRhs * s->elemSize + 16), "hipMalloc(host staging)");
    if (!rc) rc = hipFailS(hipMalloc(&s->dHostY, (size_t)big * s->maxRhs * s->elemSize + 16), "hipMalloc(host staging)");
  }
  s->timing = 1;
  // every allocation a step could need happens here: a step that fails on ONE rank after the others have enqueued
  // their half of the collective would leave them waiting forever
  if (!rc) rc = bfhipOperatorReserveRhs(op, s->maxRhs);
  if (!rc) rc = hipFailS(hipEventCreate(&s->e0), "hipEventCreate");
  if (!rc) rc = hipFailS(hipEventCreate(&s->e1), "hipEventCreate");
  if (!rc) rc = hipFailS(hipEventCreate(&s->e2), "hipEventCreate");
  if (prev >= 0) (void)hipSetDevice(prev);
  if (rc) { bfhipShardedFree(&s); return rc; }
  *out = s;
  return 0;
}

int bfhipShardedApplyDevice(BfhipSharded *s, void const *dX, size_t nrhs, void *dY, void *streamV) {
  if (!s || !dX || !dY) return bfhipFail(BFABI_ERROR_INVALID_ARGUMENTS, "NULL argument");
  if (!nrhs || nrhs > s->maxRhs) return bfhipFail(BFABI_ERROR_INVALID_ARGUMENTS, "nrhs %zu exceeds the %u this sharded apply was created for", nrhs, s->maxRhs);
  if (s->comm->aborted) return bfhipFail(BFABI_ERROR_RUNTIME_ERROR, "the communicator was aborted after a failed step on this rank");
  hipStream_t stream = (hipStream_t)streamV;
  int prev = -1;
  (void)hipGetDevice(&prev);
  int rc = prev != s->device ? hipFailS(hipSetDevice(s->device), "hipSetDevice") : 0;
  if (rc) return rc;
  ncclDataType_t const dt = s->dtype == BFHIP_F32 ? ncclFloat32 : ncclFloat64;
  size_t const scalarsPerElem = s->dtype == BFHIP_C128 ? 2 : 1;
  if (s->timing) (void)hipEventRecord(s->e0, stream);
  if (s->mode == BFHIP_SHARD_BLOCKS) {
    rc = bfhipApplyDevice(s->op, dX, nrhs, dY, stream);
    if (s->timing) (void)hipEventRecord(s->e1, stream);
    // partial results add up: the (row, col) blocks of one block row sit on different ranks, so rows receive several
    // non-zero partial sums and the order RCCL adds them in is its own -- equal to the one-GPU result to rounding
    // (1e-14 relative), not bit for bit; "rows" mode is the bit-identical one
    if (!rc) rc = ncclFail(g.AllReduce(dY, dY, (size_t)s->numRowsGlobal * nrhs * scalarsPerElem, dt, ncclSum, s->comm->comm, stream), "ncclAllReduce");
  } else {
    size_t const rowBytes = nrhs * s->elemSize;
    char *slot = (char *)s->dGather + (size_t)s->comm->rank * s->maxRows * rowBytes;
    rc = bfhipApplyDevice(s->op, dX, nrhs, slot, stream);
    if (s->timing) (void)hipEventRecord(s->e1, stream);
    if (!rc) rc = ncclFail(g.AllGather(slot, s->dGather, (size_t)s->maxRows * nrhs * scalarsPerElem, dt, s->comm->comm, stream), "ncclAllGather");
    if (!rc && s->numGroups) {
      // some ranges carry several ranks' partials: add them, in list order, into y
      uint64_t const scalarsPerRow = nrhs * scalarsPerElem;
      uint64_t const total = s->numRowsGlobal * scalarsPerRow;
      uint32_t const grid = (uint32_t)((total + 255) / 256);
      if (grid) {
        if (s->dtype == BFHIP_F32) hipLaunchKernelGGL(bfSumSegmentsKernel<float>, dim3(grid), dim3(256), 0, stream, s->dGroups, s->numGroups, s->dSrcOff, s->numRowsGlobal, (float const *)s->dGather, (float *)dY, scalarsPerRow);
        else hipLaunchKernelGGL(bfSumSegmentsKernel<double>, dim3(grid), dim3(256), 0, stream, s->dGroups, s->numGroups, s->dSrcOff, s->numRowsGlobal, (double const *)s->dGather, (double *)dY, scalarsPerRow);
        rc = hipFailS(hipGetLastError(), "segment sum launch");
      }
    } else if (!rc && s->numSegs) {
      // unit = the largest power of two <= 16 bytes dividing a row; buffers are 16-byte aligned
      size_t unit = 16;
      while (rowBytes % unit) unit /= 2;
      uint64_t const unitsPerRow = rowBytes / unit;
      uint64_t const total = s->numRowsGlobal * unitsPerRow;
      uint32_t const grid = (uint32_t)((total + 255) / 256);
      if (grid) {
        if (unit == 16) hipLaunchKernelGGL(bfScatterSegmentsKernel<uint4>, dim3(grid), dim3(256), 0, stream, s->dSegs, s->numSegs, s->numRowsGlobal, (uint4 const *)s->dGather, (uint4 *)dY, unitsPerRow);
        else if (unit == 8) hipLaunchKernelGGL(bfScatterSegmentsKernel<uint2>, dim3(grid), dim3(256), 0, stream, s->dSegs, s->numSegs, s->numRowsGlobal, (uint2 const *)s->dGather, (uint2 *)dY, unitsPerRow);
        else hipLaunchKernelGGL(bfScatterSegmentsKernel<uint32_t>, dim3(grid), dim3(256), 0, stream, s->dSegs, s->numSegs, s->numRowsGlobal, (uint32_t const *)s->dGather, (uint32_t *)dY, unitsPerRow);
        rc = hipFailS(hipGetLastError(), "segment scatter launch");
      }
    }
  }
  if (rc && !s->comm->aborted && s->comm->nranks > 1) {
    // the local stages or the collective failed to enqueue on THIS rank (arguments and allocations were settled
    // at create time, so this is a launch / runtime failure): the other ranks may already be inside the collective.
    // Abort the communicator -- their pending operation returns with an error instead of hanging -- and refuse
    // further steps on it.
    (void)g.CommAbort(s->comm->comm);
    s->comm->aborted = 1;
  }
  if (s->timing) (void)hipEventRecord(s->e2, stream);
  s->timed = !rc && s->timing;
  if (prev >= 0 && prev != s->device) (void)hipSetDevice(prev);
  return rc;
}

// z = A^T v: this rank's partial A_r^T v_r, then ONE all-reduce (see include/bfhip.h)
int bfhipShardedApplyTransposeDevice(BfhipSharded *s, void const *dV, size_t nrhs, void *dZ, void *streamV) {
  if (!s || !dV || !dZ) return bfhipFail(BFABI_ERROR_INVALID_ARGUMENTS, "NULL argument");
  if (!s->hasAdjoint) return bfhipFail(BFABI_ERROR_INVALID_ARGUMENTS, "the sharded operator was not compiled with BFHIP_FLAG_ADJOINT");
  if (!nrhs || nrhs > s->maxRhs) return bfhipFail(BFABI_ERROR_INVALID_ARGUMENTS, "nrhs %zu exceeds the %u this sharded apply was created for", nrhs, s->maxRhs);
  if (s->comm->aborted) return bfhipFail(BFABI_ERROR_RUNTIME_ERROR, "the communicator was aborted after a failed step on this rank");
  hipStream_t stream = (hipStream_t)streamV;
  int prev = -1;
  (void)hipGetDevice(&prev);
  int rc = prev != s->device ? hipFailS(hipSetDevice(s->device), "hipSetDevice") : 0;
  if (rc) return rc;
  ncclDataType_t const dt = s->dtype == BFHIP_F32 ? ncclFloat32 : ncclFloat64;
  size_t const scalarsPerElem = s->dtype == BFHIP_C128 ? 2 : 1;
  size_t const rowBytes = nrhs * s->elemSize;
  void const *vr = dV;                                   // blocks mode: the local operator has all rows
  if (s->mode == BFHIP_SHARD_ROWS) {
    if (s->numMySegs == 1 && !s->dMySegs) vr = (char const *)dV + (size_t)s->myFirstGlobal * rowBytes;
    else if (s->numMySegs) {
      size_t unit = 16;
      while (rowBytes % unit) unit /= 2;
      uint64_t const unitsPerRow = rowBytes / unit, total = s->myRows * unitsPerRow;
      uint32_t const grid = (uint32_t)((total + 255) / 256);
      if (grid) {
        if (unit == 16) hipLaunchKernelGGL(bfGatherSegmentsKernel<uint4>, dim3(grid), dim3(256), 0, stream, s->dMySegs, s->numMySegs, s->myRows, (uint4 const *)dV, (uint4 *)s->dVr, unitsPerRow);
        else if (unit == 8) hipLaunchKernelGGL(bfGatherSegmentsKernel<uint2>, dim3(grid), dim3(256), 0, stream, s->dMySegs, s->numMySegs, s->myRows, (uint2 const *)dV, (uint2 *)s->dVr, unitsPerRow);
        else hipLaunchKernelGGL(bfGatherSegmentsKernel<uint32_t>, dim3(grid), dim3(256), 0, stream, s->dMySegs, s->numMySegs, s->myRows, (uint32_t const *)dV, (uint32_t *)s->dVr, unitsPerRow);
        rc = hipFailS(hipGetLastError(), "segment gather launch");
      }
      vr = s->dVr;
    }
  }
  if (!rc) {
    if (s->mode == BFHIP_SHARD_ROWS && !s->numMySegs) rc = hipFailS(hipMemsetAsync(dZ, 0, (size_t)s->numCols * rowBytes, stream), "hipMemsetAsync");      // a rank without rows contributes zeros
    else rc = bfhipApplyTransposeDevice(s->op, vr, nrhs, dZ, stream);
  }
  if (!rc) rc = ncclFail(g.AllReduce(dZ, dZ, (size_t)s->numCols * nrhs * scalarsPerElem, dt, ncclSum, s->comm->comm, stream), "ncclAllReduce");
  if (rc && !s->comm->aborted && s->comm->nranks > 1) { (void)g.CommAbort(s->comm->comm); s->comm->aborted = 1; }
  if (prev >= 0 && prev != s->device) (void)hipSetDevice(prev);
  return rc;
}

int bfhipShardedCovMatvecDevice(BfhipSharded *s, void const *dGammaLam, uint64_t const *dRowPerm, uint64_t const *dRevRowPerm, void const *dV, void *dZ, void *stream) {
  if (!s || !dV || !dZ) return bfhipFail(BFABI_ERROR_INVALID_ARGUMENTS, "NULL argument");
  if (s->dtype == BFHIP_C128) return bfhipFail(BFABI_ERROR_TYPE_ERROR, "covariance products are defined for real operators");
  if (!s->hasAdjoint || !s->dCov) return bfhipFail(BFABI_ERROR_INVALID_ARGUMENTS, "the sharded operator was not compiled with BFHIP_FLAG_ADJOINT");
  int prev = -1;
  (void)hipGetDevice(&prev);
  int rc = prev != s->device ? hipFailS(hipSetDevice(s->device), "hipSetDevice") : 0;
  if (rc) return rc;
  uint64_t const m = s->numRowsGlobal, n = s->numCols, big = m > n ? m : n;
  char *t0 = (char *)s->dCov, *t1 = (char *)s->dCov + big * s->elemSize;
  void const *vin = dV;
  if (dRevRowPerm) { rc = bfdevScalePermute(t0, dV, NULL, 0, dRevRowPerm, m, s->dtype, stream); vin = t0; }
  if (!rc) rc = bfhipShardedApplyTransposeDevice(s, vin, 1, t1, stream);                          /* Phi^T v, every rank */
  if (!rc && dGammaLam) rc = bfdevScalePermute(t1, t1, dGammaLam, 2, NULL, n, s->dtype, stream);   /* GammaLam twice */
  if (!rc) rc = bfhipShardedApplyDevice(s, t1, 1, dRowPerm ? (void *)t0 : dZ, stream);
  if (!rc && dRowPerm) rc = bfdevScalePermute(dZ, t0, NULL, 0, dRowPerm, m, s->dtype, stream);
  if (prev >= 0 && prev != s->device) (void)hipSetDevice(prev);
  return rc;
}

static int shardedMatvec(void *ctx, void const *dX, size_t nrhs, void *dY, void *stream) { return bfhipShardedApplyDevice((BfhipSharded *)ctx, dX, nrhs, dY, stream); }

int bfhipShardedSolveGMRESDevice(BfhipSharded *s, BfhipGmresOptions const *opt, void const *dB, size_t nrhs, void const *dX0,
                                 size_t *numIter, double *residual, void *dX, void *stream) {
  if (!s) return bfhipFail(BFABI_ERROR_INVALID_ARGUMENTS, "NULL sharded operator");
  if (s->dtype != BFHIP_C128) return bfhipFail(BFABI_ERROR_NOT_IMPLEMENTED, "GMRES is implemented for complex operators");
  if (s->numRowsGlobal != s->numCols) return bfhipFail(BFABI_ERROR_INVALID_ARGUMENTS, "GMRES needs a square operator (linalg.c:85-87)");
  if (!nrhs || nrhs > s->maxRhs) return bfhipFail(BFABI_ERROR_INVALID_ARGUMENTS, "nrhs %zu exceeds the %u this sharded apply was created for", nrhs, s->maxRhs);
  int const timing = s->timing;
  s->timing = 0;                 // no event pairs between the iterations' launches
  int const rc = bfGmresSolve(shardedMatvec, s, s->numRowsGlobal, s->device, opt, dB, nrhs, dX0, numIter, residual, dX, stream);
  s->timing = timing;
  return rc;
}

size_t bfhipShardedGetNumRows(BfhipSharded const *s) { return s ? s->numRowsGlobal : 0; }
size_t bfhipShardedGetNumCols(BfhipSharded const *s) { return s ? s->numCols : 0; }
BfhipOperator *bfhipShardedOperator(BfhipSharded const *s) { return s ? s->op : NULL; }

// host vectors through the sharded step (the vtable shim): packed, double precision on the host like bfhipApply
int bfhipShardedApplyHost(BfhipSharded *s, int transpose, void const *X, size_t ldx, size_t nrhs, void *Y, size_t ldy) {
  if (!s || !X || !Y) return bfhipFail(BFABI_ERROR_INVALID_ARGUMENTS, "NULL argument");
  if (!nrhs || nrhs > s->maxRhs || ldx < nrhs || ldy < nrhs) return bfhipFail(BFABI_ERROR_INVALID_ARGUMENTS, "bad nrhs / leading dimension (the sharded apply was created for %u right-hand sides)", s->maxRhs);
  int prev = -1;
  (void)hipGetDevice(&prev);
  int rc = prev != s->device ? hipFailS(hipSetDevice(s->device), "hipSetDevice") : 0;
  if (rc) return rc;
  uint64_t const n = transpose ? s->numRowsGlobal : s->numCols, m = transpose ? s->numCols : s->numRowsGlobal;
  size_t const es = s->elemSize, hostEs = s->dtype == BFHIP_C128 ? 16 : 8;
  void *hx = malloc((n * nrhs ? n * nrhs : 1) * es), *hy = malloc((m * nrhs ? m * nrhs : 1) * es);
  if (!hx || !hy) { free(hx); free(hy); if (prev >= 0 && prev != s->device) (void)hipSetDevice(prev); return bfhipFail(BFABI_ERROR_MEMORY_ERROR, "host OOM"); }
  if (es == hostEs) for (uint64_t i = 0; i < n; ++i) memcpy((char *)hx + i * nrhs * es, (char const *)X + i * ldx * es, nrhs * es);
  else for (uint64_t i = 0; i < n; ++i) for (size_t q = 0; q < nrhs; ++q) ((float *)hx)[i * nrhs + q] = (float)((double const *)X)[i * ldx + q];
  rc = hipFailS(hipMemcpy(s->dHostX, hx, n * nrhs * es, hipMemcpyHostToDevice), "hipMemcpy H2D");
  if (!rc) rc = transpose ? bfhipShardedApplyTransposeDevice(s, s->dHostX, nrhs, s->dHostY, NULL) : bfhipShardedApplyDevice(s, s->dHostX, nrhs, s->dHostY, NULL);
  if (!rc) rc = hipFailS(hipMemcpy(hy, s->dHostY, m * nrhs * es, hipMemcpyDeviceToHost), "hipMemcpy D2H");
  if (!rc) {
    if (es == hostEs) for (uint64_t i = 0; i < m; ++i) memcpy((char *)Y + i * ldy * es, (char *)hy + i * nrhs * es, nrhs * es);
    else for (uint64_t i = 0; i < m; ++i) for (size_t q = 0; q < nrhs; ++q) ((double *)Y)[i * ldy + q] = ((float *)hy)[i * nrhs + q];
  }
  free(hx); free(hy);
  if (prev >= 0 && prev != s->device) (void)hipSetDevice(prev);
  return rc;
}

int bfhipShardedLastTimes(BfhipSharded *s, double *localMs, double *collectiveMs) {
  if (!s || !s->timed) return bfhipFail(BFABI_ERROR_INVALID_ARGUMENTS, "the last sharded apply was not timed (none ran, it failed, or bfhipShardedSetTiming(0))");
  int rc = hipFailS(hipEventSynchronize(s->e2), "hipEventSynchronize");
  float a = 0, b = 0;
  if (!rc) rc = hipFailS(hipEventElapsedTime(&a, s->e0, s->e1), "hipEventElapsedTime");
  if (!rc) rc = hipFailS(hipEventElapsedTime(&b, s->e1, s->e2), "hipEventElapsedTime");
  if (localMs) *localMs = a;
  if (collectiveMs) *collectiveMs = b;
  return rc;
}

int bfhipShardedSetTiming(BfhipSharded *s, int enabled) {
  if (!s) return bfhipFail(BFABI_ERROR_INVALID_ARGUMENTS, "NULL argument");
  s->timing = enabled != 0;
  if (!s->timing) s->timed = 0;
  return 0;
}

}  // extern "C"
