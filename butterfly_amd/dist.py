"""Row-sharded apply over the GPUs of one node.

The top level of a fac_helm2 operator is a dense grid of blocks over the
level-2 target nodes (reference src/fac_helm2.c:956-987), and
bfMatBlockDenseMul computes every block row independently from the full x
(src/mat_block_dense.c:534-566): no butterfly stage crosses a top-level row
block.  So each rank (one process per GPU) compiles only the block rows it
owns, keeps a replica of x, and a step is

    y_local = A[rows of this rank, :] @ x          (all stages, on this GPU)
    y = all_gather(y_local) reordered to row order  (ONE RCCL collective, xGMI)

No reduction is needed: outputs are disjoint.  Block rows are dealt to ranks
by longest-processing-time-first on leaf bytes (12 non-empty row blocks on a
circle => 8 ranks cannot do better than 2/12 of the work on the busiest one:
6.17x at N = 262144).

Where that bound bites (8 ranks), the finer "blocks" mode deals the 144
top-level (row, col) blocks instead (LPT again: 7.98x ideal at N = 262144);
every rank then produces a full-length partial y and the step ends with ONE
all-reduce (sum) of y instead of the all-gather.  `choose_mode` picks rows
unless blocks is better balanced by more than 2 %.
"""
from __future__ import annotations

import numpy as np

from . import helm2_structure as hs


def row_block_weights(desc):
    """Leaf elements under each top-level block row."""
    w = [0] * len(desc.meta["top_rows"])
    if hasattr(desc, "subtree_leaf_elems"):                      # array-backed (native layout)
        tot = desc.subtree_leaf_elems()
        for (c, _, _), rb in zip(desc.children[desc.root], desc.top_row_block):
            w[rb] += int(tot[c])
        return w
    kind = np.asarray(desc.kind)
    own = np.asarray(desc.rows, dtype=np.int64) * np.asarray(desc.cols, dtype=np.int64) * (kind == hs.NODE_DENSE)
    for (c, _, _), rb in zip(desc.children[desc.root], desc.top_row_block):
        tot, stack = 0, [c]
        while stack:
            v = stack.pop()
            tot += int(own[v])
            stack.extend(ch for ch, _, _ in desc.children[v])
        w[rb] += tot
    return w


def assign_row_blocks(weights, world):
    """LPT bin packing; returns (owner per row block, load per rank)."""
    order = sorted(range(len(weights)), key=lambda i: (-weights[i], i))
    loads = [0] * world
    owner = [0] * len(weights)
    for rb in order:
        r = min(range(world), key=lambda q: (loads[q], q))
        owner[rb] = r
        loads[r] += weights[rb]
    return owner, loads


def block_weights(desc):
    """Leaf elements under each top-level (row, col) block, in child order."""
    if hasattr(desc, "subtree_leaf_elems"):
        tot = desc.subtree_leaf_elems()
        return [int(tot[c]) for (c, _, _) in desc.children[desc.root]]
    kind = np.asarray(desc.kind)
    own = np.asarray(desc.rows, dtype=np.int64) * np.asarray(desc.cols, dtype=np.int64) * (kind == hs.NODE_DENSE)
    w = []
    for (c, _, _) in desc.children[desc.root]:
        tot, stack = 0, [c]
        while stack:
            v = stack.pop()
            tot += int(own[v])
            stack.extend(ch for ch, _, _ in desc.children[v])
        w.append(tot)
    return w


def row_partition(desc, world, root=None):
    """bfhipRowPartition: balanced contiguous row ranges (cuts[world + 1], leaf elements kept per rank).  Cuts fall on
    clean positions only (no leaf writing y straddles one), one level or more below the top-level row blocks: the
    level-3+ rows inside each level-2 BlockDense (reference src/fac_helm2.c:814-858) and the target-node boundaries of
    the evaluation factors, with the source-side factors a range needs replicated."""
    import ctypes as C

    from . import _capi
    lib = _capi.load()
    da = _capi.DescArrays(desc, root=root)
    cuts = np.zeros(world + 1, dtype=np.uint64)
    loads = np.zeros(world, dtype=np.uint64)
    _capi.check(lib.bfhipRowPartition(da.byref(), world, cuts.ctypes.data, loads.ctypes.data))
    return [int(c) for c in cuts], [int(v) for v in loads]


def rowsum_partition(desc, world):
    """Whole block rows where they fit, a column share of a block row where they do not: every rank gets whole top-level
    block rows (largest first, least-loaded rank) as long as its load stays within the mean; each remaining block row
    is split by COLUMNS -- its (row, col) blocks dealt, largest first, to the k least-loaded ranks, k = 2 unless that
    leaves the busiest rank more than 3 % over the mean.  No leaf is replicated (unlike row ranges) and the loads match
    to the size of the smallest blocks; the price is that a shared block row's result is the sum of k partials, added in
    rank order after the one all-gather (deterministic; equal to one GPU to rounding, not bit for bit).
    Returns (owner per top-level child, load per rank, segments [(block row, rank)] in list order)."""
    bw = block_weights(desc)
    trb = list(desc.top_row_block)
    nrows = len(desc.meta["top_rows"])
    total = sum(bw)
    row_w = [0] * nrows
    for c, rb in enumerate(trb):
        row_w[rb] += bw[c]
    best = None
    # which block rows stay whole decides how even the loads can get: 4 rows of 6.23 GB and 8 of 5.75 GB on 8 ranks
    # (N = 262144) balance to 1e-3 when the LARGE rows are the shared ones (5.75 + 6.23 / 2 on every rank) and only to 3 %
    # when they are the whole ones -- both orders are tried, the better one kept
    for k, big_first in ((2, True), (2, False), (3, True), (3, False), (4, True), (world, True)):
        k = min(k, world)
        loads = [0] * world
        owner = [-1] * len(bw)
        whole, split = {}, []
        for rb in sorted(range(nrows), key=lambda i: ((-row_w[i]) if big_first else row_w[i], i)):
            q = min(range(world), key=lambda r: (loads[r], r))
            if loads[q] + row_w[rb] <= total / world * 1.0001:
                whole[rb] = q
                loads[q] += row_w[rb]
            else:
                split.append(rb)
        for c, rb in enumerate(trb):
            if rb in whole:
                owner[c] = whole[rb]
        for rb in split:
            cand = sorted(range(world), key=lambda r: (loads[r], r))[:k]
            cols = sorted((c for c in range(len(bw)) if trb[c] == rb), key=lambda c: (-bw[c], c))
            if k == 2 and len(cols) <= 16:
                # two ranks, at most 16 blocks: the best of all 2^n column splits (the larger of the two resulting loads)
                a, b = cand
                best_mask, best_val = 0, None
                for mask in range(1 << len(cols)):
                    wa = sum(bw[c] for i, c in enumerate(cols) if mask >> i & 1)
                    val = max(loads[a] + wa, loads[b] + row_w[rb] - wa)
                    if best_val is None or val < best_val:
                        best_mask, best_val = mask, val
                for i, c in enumerate(cols):
                    q = a if best_mask >> i & 1 else b
                    owner[c] = q
                    loads[q] += bw[c]
                continue
            for c in cols:
                q = min(cand, key=lambda r: (loads[r], r))
                owner[c] = q
                loads[q] += bw[c]
        if best is None or max(loads) < best[1]:
            best = (owner, max(loads), loads)
        if max(loads) <= 1.002 * total / world:
            break
    owner, _, loads = best
    segs = []
    for rb in range(nrows):
        for r in sorted({owner[c] for c in range(len(bw)) if trb[c] == rb}):
            segs.append((rb, r))
    return owner, loads, segs


def choose_mode(desc, world, requested="auto"):
    """How the operator is dealt to `world` ranks:
    'rows'      contiguous row RANGES from row_partition (balanced to a few % at any world size; ONE all-gather;
                bit-identical to one GPU) -- the default wherever it replicates nothing;
    'rowblocks' whole top-level block rows by LPT (round 2's "rows": 12 blocks on a circle bound 8 ranks at 6.2x);
    'blocks'    top-level (row, col) blocks by LPT + ONE all-reduce (equal to one GPU to rounding only);
    'rowsum'    whole block rows + column shares of the rest (rowsum_partition): no replication, balanced to ~3 %, ONE
                all-gather, the partials of a shared row added in rank order afterwards (deterministic; equal to one
                GPU to rounding) -- the default where row ranges would replicate leaves (8 ranks on 12 block rows:
                measured 1.55 ms per shard against 1.67 ms at N = 262144)."""
    if requested in ("rows", "rowblocks", "blocks", "rowsum"):
        return requested
    if world == 1:
        return "rows"
    # row ranges are free where the top-level block rows divide evenly (2 and 4 ranks on a closed curve: no leaf is
    # replicated); where a block row has to be shared (8 ranks: +6 % leaves for the replicated first-applied factors)
    # the column split of rowsum is the cheaper way to share it
    has_blocks = getattr(desc, "top_row_block", None) is not None
    from . import _capi
    try:
        cuts, loads = row_partition(desc, world)
    except _capi.BfhipError as e:
        # fewer clean cut positions than ranks (small or tall-leaf operands: bfhipRowPartition answers INVALID_ARGUMENTS): no row
        # ranges -- share block rows instead.  Anything else (a malformed descriptor, a bug) propagates: a silent change of the
        # sharding mode on ONE rank would leave the ranks in different collectives.
        if e.code != 1 or not has_blocks:
            raise
        return "rowsum"
    total = int(desc.subtree_leaf_elems()[desc.root]) if hasattr(desc, "subtree_leaf_elems") else sum(row_block_weights(desc))
    return "rows" if sum(loads) <= 1.005 * total or not has_blocks else "rowsum"


class ShardLayout:
    """Who owns which rows, and where each global row lands in the gathered
    (rank-major, padded) buffer."""

    def __init__(self, top_rows, owner, world, segments=None):
        """segments (rowsum): [(block row, rank)] in list order -- a block row may appear under several ranks, each
        contributing a partial result for it; top_rows / owner then describe the block rows (owner unused)."""
        self.top_rows = list(top_rows)
        self.owner = list(owner)
        self.world = world
        self.row_offsets = np.concatenate([[0], np.cumsum(self.top_rows)]).astype(np.int64)
        self.n = int(self.row_offsets[-1])
        self.segments = None
        if segments is not None:
            self.segments = [(int(rb), int(r)) for rb, r in segments]
            self.blocks_of = [[rb for rb, r in self.segments if r == q] for q in range(world)]
            self.rows_of = [sum(self.top_rows[rb] for rb in blks) for blks in self.blocks_of]
            self.max_rows = max(self.rows_of) if self.rows_of else 0
            pos = [q * self.max_rows for q in range(world)]
            self.seg_src = []                      # gather-buffer row of each segment's first row
            for rb, r in self.segments:
                self.seg_src.append(pos[r])
                pos[r] += self.top_rows[rb]
            self.gather_index = None
            return
        self.blocks_of = [[rb for rb in range(len(owner)) if owner[rb] == r] for r in range(world)]
        self.rows_of = [sum(self.top_rows[rb] for rb in blks) for blks in self.blocks_of]
        self.max_rows = max(self.rows_of) if self.rows_of else 0
        idx = np.empty(self.n, dtype=np.int64)
        for r in range(world):
            pos = r * self.max_rows
            for rb in self.blocks_of[r]:
                m = self.top_rows[rb]
                idx[self.row_offsets[rb]:self.row_offsets[rb] + m] = np.arange(pos, pos + m)
                pos += m
        self.gather_index = idx


class ShardedApply:
    """y = A x with A's top-level block rows spread over the ranks of a
    torch.distributed group.  `local_apply(x, out)` runs this rank's rows
    (HipOperator.apply_device on a GPU; tests inject a CPU stand-in)."""

    def __init__(self, layout: ShardLayout, rank, local_apply, device, dtype, nrhs=1, group=None, mode="rows",
                 force_collective=False):
        import torch
        self.layout, self.rank, self.local_apply, self.group, self.mode = layout, rank, local_apply, group, mode
        self.force_collective = force_collective   # run the collective even on one rank (rehearsal)
        tail = () if nrhs == 1 else (nrhs,)
        if mode == "blocks":
            self.local_rows = layout.n
            self.y_local = torch.empty((layout.n,) + tail, dtype=dtype, device=device)
            return
        self.local_rows = layout.rows_of[rank]
        self.y_local = torch.empty((self.local_rows,) + tail, dtype=dtype, device=device)
        self.pad = torch.zeros((layout.max_rows,) + tail, dtype=dtype, device=device)
        self.gathered = torch.empty((layout.world * layout.max_rows,) + tail, dtype=dtype, device=device)
        self.index = torch.from_numpy(layout.gather_index).to(device) if layout.gather_index is not None else None

    def my_rows(self):
        """Global row indices of this rank's entries of a length-n vector, in the order its local operator yields them."""
        lay = self.layout
        if self.mode == "blocks":
            return np.arange(lay.n, dtype=np.int64)
        blks = lay.blocks_of[self.rank]
        if not blks:
            return np.zeros(0, dtype=np.int64)
        return np.concatenate([np.arange(lay.row_offsets[rb], lay.row_offsets[rb] + lay.top_rows[rb]) for rb in blks]).astype(np.int64)

    def apply_transpose(self, v, local_apply_transpose, num_cols):
        """z = A^T v with v replicated: this rank applies A_r^T (local_apply_transpose(v_r, out)) to ITS entries of v and
        the full-length partials add up -- ONE all-reduce (bfhipShardedApplyTransposeDevice is the C-ABI twin)."""
        import torch
        import torch.distributed as dist
        if getattr(self, "_rows_idx", None) is None:
            self._rows_idx = torch.from_numpy(self.my_rows()).to(v.device)
        vr = v.index_select(0, self._rows_idx)
        z = torch.zeros((num_cols,) + tuple(v.shape[1:]), dtype=v.dtype, device=v.device)
        if vr.shape[0]:
            local_apply_transpose(vr, z)
        if self.layout.world > 1 or self.force_collective:
            buf = torch.view_as_real(z) if z.is_complex() else z
            dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group)
        return z

    def cov_matvec(self, local_apply_transpose, num_cols, gamma, row_perm, rev_row_perm, v):
        """cov_matvec of examples/covariance/lbo_cov.c:48-60 over the sharded operator: z = P A G G A^T P' v with the
        reference's scatter-order permutations (out[perm[i]] = in[i], src/vec_real.c:312-329); every vector replicated."""
        import torch
        t = v
        if rev_row_perm is not None:
            t = torch.empty_like(v); t[rev_row_perm] = v
        u = self.apply_transpose(t, local_apply_transpose, num_cols)
        if gamma is not None:
            u = u * gamma * gamma
        z = self(u)
        if row_perm is not None:
            zz = torch.empty_like(z); zz[row_perm] = z
            z = zz
        return z

    def __call__(self, x):
        import torch.distributed as dist
        import torch
        self.local_apply(x, self.y_local)
        if self.layout.world == 1 and not self.force_collective:
            return self.y_local
        if self.mode == "blocks":
            # partial results add up: one all-reduce (RCCL) replaces the all-gather
            buf = torch.view_as_real(self.y_local) if self.y_local.is_complex() else self.y_local
            dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group)
            return self.y_local
        self.pad[:self.local_rows] = self.y_local
        # collectives run on the real view (re, im pairs): every backend moves doubles
        src = torch.view_as_real(self.pad) if self.pad.is_complex() else self.pad
        dst = torch.view_as_real(self.gathered) if self.gathered.is_complex() else self.gathered
        dist.all_gather_into_tensor(dst, src, group=self.group)
        if self.layout.segments is not None:
            # rowsum: the partials of a block row, in list (rank) order
            y = torch.zeros((self.layout.n,) + tuple(self.gathered.shape[1:]), dtype=self.gathered.dtype, device=self.gathered.device)
            seen = set()
            for (rb, r), s0 in zip(self.layout.segments, self.layout.seg_src):
                g0, m = int(self.layout.row_offsets[rb]), self.layout.top_rows[rb]
                if rb in seen:
                    y[g0:g0 + m] += self.gathered[s0:s0 + m]
                else:
                    y[g0:g0 + m] = self.gathered[s0:s0 + m]
                    seen.add(rb)
            return y
        return self.gathered.index_select(0, self.index)


class RcclShardedApply:
    """The same step through the C-ABI (include/bfhip.h "multi-GPU"): the local stages and the ONE RCCL
    collective -- in-place ncclAllGather + segment reordering for "rows", ncclAllReduce for "blocks" --
    are enqueued by libbfhip.so on the caller's stream; Python only ships the 128-byte communicator id
    from rank 0 to the others (through `bcast`, e.g. a torch.distributed object broadcast).  This is what
    bench.py runs on GPUs; `ShardedApply` above is the torch.distributed rendition the CPU (gloo) tests
    drive with a stand-in local apply."""

    def __init__(self, layout: ShardLayout, rank, op, device_index, nrhs=1, mode="rows", bcast=None):
        import ctypes as C

        from . import _capi
        self._lib = lib = _capi.load()
        self.layout, self.rank, self.op, self.mode, self.nrhs = layout, rank, op, mode, nrhs
        ident = C.create_string_buffer(128)
        if rank == 0:
            _capi.check(lib.bfhipCommGetUniqueId(ident))
        if layout.world > 1:
            if bcast is None:
                raise ValueError("more than one rank needs a way to ship the communicator id")
            ident = C.create_string_buffer(bcast(ident.raw if rank == 0 else None), 128)
        self._comm = C.c_void_p()
        _capi.check(lib.bfhipCommInitRank(ident, layout.world, rank, device_index, C.byref(self._comm)))
        spec = _capi.BfhipShardSpec()
        spec.structSize = C.sizeof(spec)
        spec.mode = _capi.SHARD_BLOCKS if mode == "blocks" else _capi.SHARD_ROWS      # "rows" and "rowblocks": all-gather of row segments
        spec.numRowsGlobal = layout.n
        if layout.segments is not None:
            self._seg_rows = np.ascontiguousarray([layout.top_rows[rb] for rb, _ in layout.segments], dtype=np.uint64)
            self._seg_owner = np.ascontiguousarray([r for _, r in layout.segments], dtype=np.uint32)
            self._seg_off = np.ascontiguousarray([layout.row_offsets[rb] for rb, _ in layout.segments], dtype=np.uint64)
            spec.segGlobalOff = self._seg_off.ctypes.data
        else:
            self._seg_rows = np.ascontiguousarray(layout.top_rows, dtype=np.uint64)
            self._seg_owner = np.ascontiguousarray(layout.owner, dtype=np.uint32)
        if mode != "blocks":
            spec.numSegments = len(self._seg_rows)
            spec.segRows, spec.segOwner = self._seg_rows.ctypes.data, self._seg_owner.ctypes.data
        self._sh = C.c_void_p()
        _capi.check(lib.bfhipShardedCreate(op.handle, self._comm, C.byref(spec), nrhs, C.byref(self._sh)))
        self._y = None

    def __call__(self, x):
        import ctypes as C

        import torch

        from . import _capi
        if self._y is None:
            shape = (self.layout.n,) if x.dim() == 1 else (self.layout.n, x.shape[1])
            self._y = torch.empty(shape, dtype=x.dtype, device=x.device)
        nrhs = 1 if x.dim() == 1 else x.shape[1]
        s = torch.cuda.current_stream(x.device)
        _capi.check(self._lib.bfhipShardedApplyDevice(self._sh, C.c_void_p(x.data_ptr()), nrhs, C.c_void_p(self._y.data_ptr()),
                                                      C.c_void_p(s.cuda_stream)))
        return self._y

    def apply_transpose(self, v):
        """z = A^T v on every rank (v replicated): bfhipShardedApplyTransposeDevice -- this rank's partial A_r^T v_r, ONE all-reduce."""
        import ctypes as C

        import torch

        from . import _capi
        ncols = int(self._lib.bfhipShardedGetNumCols(self._sh))
        z = torch.empty((ncols,) + tuple(v.shape[1:]), dtype=v.dtype, device=v.device)
        nrhs = 1 if v.dim() == 1 else v.shape[1]
        s = torch.cuda.current_stream(v.device)
        _capi.check(self._lib.bfhipShardedApplyTransposeDevice(self._sh, C.c_void_p(v.data_ptr()), nrhs, C.c_void_p(z.data_ptr()), C.c_void_p(s.cuda_stream)))
        return z

    def cov_matvec(self, gamma, row_perm, rev_row_perm, v):
        """bfhipShardedCovMatvecDevice: z = P A G G A^T P' v in one call (real operators; perms int64 / uint64 device tensors or None)."""
        import ctypes as C

        import torch

        from . import _capi
        z = torch.empty(self.layout.n, dtype=v.dtype, device=v.device)
        s = torch.cuda.current_stream(v.device)
        ptr = lambda t: None if t is None else C.c_void_p(t.data_ptr())
        _capi.check(self._lib.bfhipShardedCovMatvecDevice(self._sh, ptr(gamma), ptr(row_perm), ptr(rev_row_perm), ptr(v), ptr(z), C.c_void_p(s.cuda_stream)))
        return z

    def solve_gmres(self, b, x0=None, tol=1e-12, max_num_iter=100, orthogonalization=0, precond=None):
        """bfhipShardedSolveGMRESDevice: (x, numIter, residual); b / x0 replicated on every rank."""
        import ctypes as C

        import torch

        from . import _capi
        opt = _capi.BfhipGmresOptions()
        opt.structSize, opt.orthogonalization, opt.tol, opt.maxNumIter = C.sizeof(opt), orthogonalization, tol, max_num_iter
        opt.solveM = precond.handle if precond is not None else None
        x = torch.empty_like(b)
        nrhs = 1 if b.dim() == 1 else b.shape[1]
        it, res = C.c_size_t(), C.c_double()
        s = torch.cuda.current_stream(b.device)
        _capi.check(self._lib.bfhipShardedSolveGMRESDevice(self._sh, C.byref(opt), C.c_void_p(b.data_ptr()), nrhs,
                                                           None if x0 is None else C.c_void_p(x0.data_ptr()), C.byref(it), C.byref(res),
                                                           C.c_void_p(x.data_ptr()), C.c_void_p(s.cuda_stream)))
        return x, int(it.value), float(res.value)

    def mat_new(self):
        """The BfMat vtable shim over the sharded operator (bfhipShardedMatNew); the caller deletes it through its Delete slot."""
        return self._lib.bfhipShardedMatNew(self._sh, 0)

    def set_timing(self, enabled):
        """Record (or not) the three events behind last_times(); they cost ~17 us of stream time per step."""
        from . import _capi
        _capi.check(self._lib.bfhipShardedSetTiming(self._sh, 1 if enabled else 0))

    def last_times(self):
        """(local stage ms, collective ms) of the most recent step, from hipEvents on its stream."""
        import ctypes as C

        from . import _capi
        a, b = C.c_double(), C.c_double()
        _capi.check(self._lib.bfhipShardedLastTimes(self._sh, C.byref(a), C.byref(b)))
        return a.value, b.value

    def close(self):
        import ctypes as C
        if self._sh:
            self._lib.bfhipShardedFree(C.byref(self._sh))
        if self._comm:
            self._lib.bfhipCommDestroy(C.byref(self._comm))


# --------------------------------------------------------------------------
# GMRES over a sharded apply ("with multi-GPU the
# allgathered iterate is already replicated")
# --------------------------------------------------------------------------
def _givens(a, b):
    """bfVecComplexGetGivensRotation, reference src/vec_complex.c:275-286."""
    if abs(b) == 0:
        return 1.0 + 0j, 0.0 + 0j
    if abs(b) > abs(a):
        tmp = -a / b
        sn = 1 / (1 + abs(tmp) ** 2) ** 0.5
        return tmp * sn, sn + 0j
    tmp = -b / a
    cs = 1 / (1 + abs(tmp) ** 2) ** 0.5
    return cs + 0j, tmp * cs


def _rot(col, i, c, s):
    """mulInplace_givensComplex, src/vec_complex.c:155-168, on entries (i, i + 1) of a Python list."""
    z0, z1 = col[i], col[i + 1]
    col[i], col[i + 1] = c.conjugate() * z0 - s * z1, s * z0 + c * z1


def sharded_solve_gmres(step, b, x0=None, tol=1e-12, max_num_iter=100):
    """bfSolveGMRES (reference src/linalg.c:47-317) with the matvec spread over the ranks of a
    process group: `step(x)` is a ShardedApply (or any callable) returning the FULL product on
    every rank.  All vectors are replicated -- the apply's closing collective already hands every
    rank the whole iterate -- so the Krylov recurrences run redundantly and identically on each
    GPU (O(j N) per iteration, nothing next to the apply) and the only communication of an iteration
    is the apply's one collective.  Same algorithm and quirks as the single-GPU bfhipSolveGMRES
    (unrestarted, Gram-Schmidt as two batched classical passes, residual = max_p |s_{j+1,p}| / max_p ||r_p||,
    a converged solve uses j basis vectors).  b: [n] or [n, nrhs] complex tensor on the apply's device.
    Returns (x, num_iter, residual)."""
    import torch
    one_d = b.dim() == 1
    B = b[:, None] if one_d else b
    n, nrhs = B.shape
    call = (lambda v: step(v[:, 0].contiguous())[:, None]) if one_d else (lambda v: step(v.contiguous()))
    X0 = torch.zeros_like(B) if x0 is None else (x0[:, None] if one_d else x0)
    # `step` may return its own reused buffer: clone what is kept
    R = B - call(X0)
    rnorm = torch.linalg.vector_norm(R, dim=0)
    beta = float(rnorm.max())
    # the Krylov basis as ONE tensor [max_num_iter + 1, n, nrhs]: an iteration's projections are two batched
    # contractions per Gram-Schmidt pass (CGS2: classical Gram-Schmidt, run twice -- as stable as the
    # reference's modified Gram-Schmidt, src/linalg.c:174-184, but one host synchronisation per iteration
    # instead of one per basis vector)
    V = torch.empty((max_num_iter + 1, n, nrhs), dtype=B.dtype, device=B.device)
    V[0] = R / rnorm
    S = [[complex(rnorm[p]) if i == 0 else 0j for i in range(max_num_iter + 1)] for p in range(nrhs)]
    H, J = [], {}
    residual, converged, j = float("inf"), False, 0
    for j in range(max_num_iter):
        W = call(V[j]).clone()
        Vj = V[:j + 1]
        h1 = torch.einsum("inp,np->ip", Vj.conj(), W)
        W -= torch.einsum("inp,ip->np", Vj, h1)
        h2 = torch.einsum("inp,np->ip", Vj.conj(), W)
        W -= torch.einsum("inp,ip->np", Vj, h2)
        wnorm = torch.linalg.vector_norm(W, dim=0)
        V[j + 1] = W / wnorm
        hcol = torch.cat([h1 + h2, wnorm[None].to(h1.dtype)], dim=0).tolist()     # the iteration's one host sync
        Hj = [[complex(hcol[i][p]) for i in range(j + 2)] for p in range(nrhs)]
        for p in range(nrhs):
            col = Hj[p]
            for i in range(j):                       # earlier rotations, then the new one (:206-228)
                _rot(col, i, *J[(i, p)])
            J[(j, p)] = _givens(col[j], col[j + 1])
            _rot(col, j, *J[(j, p)])
            _rot(S[p], j, *J[(j, p)])
        H.append(Hj)
        residual = max(abs(S[p][j + 1]) for p in range(nrhs)) / beta
        if residual < tol:                           # :235-241: breaks before j is incremented
            converged = True
            break
    if not converged:
        j = max_num_iter
    X = X0.clone()
    for p in range(nrhs):                            # back substitution and update (:245-285)
        y = [0j] * j
        for r in range(j - 1, -1, -1):
            acc = S[p][r] - sum(H[c][p][r] * y[c] for c in range(r + 1, j))
            y[r] = acc / H[r][p][r]
        for i in range(j):
            X[:, p] += V[i][:, p] * y[i]
    return (X[:, 0] if one_d else X), j, residual
