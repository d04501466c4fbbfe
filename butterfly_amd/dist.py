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


def choose_mode(desc, world, requested="auto"):
    """'rows' (all-gather) or 'blocks' (all-reduce) for this operand and world size."""
    if requested in ("rows", "blocks"):
        return requested
    if world == 1:
        return "rows"
    _, lr = assign_row_blocks(row_block_weights(desc), world)
    _, lb = assign_row_blocks(block_weights(desc), world)
    return "rows" if max(lr) <= 1.02 * max(lb) else "blocks"


class ShardLayout:
    """Who owns which rows, and where each global row lands in the gathered
    (rank-major, padded) buffer."""

    def __init__(self, top_rows, owner, world):
        self.top_rows = list(top_rows)
        self.owner = list(owner)
        self.world = world
        self.row_offsets = np.concatenate([[0], np.cumsum(self.top_rows)]).astype(np.int64)
        self.n = int(self.row_offsets[-1])
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
        self.index = torch.from_numpy(layout.gather_index).to(device)

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
        return self.gathered.index_select(0, self.index)
