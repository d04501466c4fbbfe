"""Test infrastructure: a numpy interpreter of the engine's flattened plan.

It executes exactly what the HIP kernels are specified to do (per item:
sum over pieces of  A_piece @ x_segment, then the reduce passes), reading the
plan through the inspection C-ABI (BFHIP_FLAG_PLAN_ONLY; no GPU needed).  It
lets the CPU test-suite check the host logic -- scheduling, row groups,
packing, reduce intervals -- against the oracle without a device.  It is NOT a
fallback for the engine and lives only under tests/."""
from __future__ import annotations

import ctypes as C

import numpy as np

from butterfly_amd import _capi

BF_ITEM_OUT_Y = 1 << 16
BF_PIECE_IN_X = 1
BF_PIECE_IDENTITY = 2
BF_PIECE_ROWMAJOR = 4
BF_ITEM_ROWMAJOR = 1 << 17
BF_ITEM_MERGED = 1 << 18
BF_ITEM_SMALL = 1 << 19
BF_ITEM_TNARROW = 1 << 20


def _view(ptr, count, dtype):
    if count == 0:
        return np.zeros(0, dtype=dtype)
    buf = (C.c_char * (count * dtype.itemsize)).from_address(ptr)
    return np.frombuffer(buf, dtype=dtype)


def run_plan(op, x, transpose=False):
    """op: butterfly_amd.operator.HipOperator compiled with FLAG_PLAN_ONLY
    (and FLAG_ADJOINT for transpose=True: the plan of A^T over the same arena)."""
    lib = _capi.load()
    info = _capi.BfhipPlanInfo()
    info.structSize = C.sizeof(info)
    _capi.check(lib.bfhipPlanGetInfo(op.handle, C.byref(info)))
    dt = {0: np.complex128, 1: np.float64, 2: np.float32}[info.dtype]
    epl = info.epl
    # BFHIP_FLAG_ADJOINT_PACKED: the adjoint plan is a FORWARD plan of the transposed expression over an arena of its own --
    # its stages are run with the forward kernels' semantics on that arena
    packed = transpose and int(info.reserved) == 1
    if packed:
        arena = np.zeros(int(info.arenaElemsT), dtype=dt)
        _capi.check(lib.bfhipPlanPackArenaT(op.handle, arena.ctypes.data))
    else:
        arena = np.zeros(int(info.arenaElems), dtype=dt)
        _capi.check(lib.bfhipPlanPackArena(op.handle, arena.ctypes.data))
    tstages, transpose = transpose, transpose and not packed      # from here on `transpose` = "transposed-kernel semantics"
    x = np.asarray(x, dtype=dt)
    one_d = x.ndim == 1
    if one_d:
        x = x[:, None]
    nrhs = x.shape[1]
    stage0 = int(info.numStages) if tstages else 0
    nstages = int(info.numStagesT) if tstages else int(info.numStages)
    if tstages:
        assert nstages > 0, "operator has no transposed plan (FLAG_ADJOINT)"
    y = np.full((int(info.numCols if tstages else info.numRows), nrhs), np.nan, dtype=dt)
    temp = np.full((int(max(info.tempElems, info.tempElemsT)), nrhs), np.nan, dtype=dt)
    for s in range(stage0, stage0 + nstages):
        sv = _capi.BfhipStageView()
        sv.structSize = C.sizeof(sv)
        _capi.check(lib.bfhipPlanGetStage(op.handle, s, C.byref(sv)))
        items = _view(sv.items, int(sv.numItems), _capi.ITEM_DTYPE)
        pieces = _view(sv.pieces, int(sv.numPieces), _capi.PIECE_DTYPE)
        if transpose:        # bfStageKernelT: at most 64 columns of A per item
            assert ((items["mrFlags"] & 0xFFFF) <= 64).all()
        narrow = (items["mrFlags"] & BF_ITEM_TNARROW) != 0
        if narrow.any():     # 16-column items of tall leaves open a transposed stage's list (their own launch)
            k = int(narrow.sum())
            assert transpose and narrow[:k].all() and ((items["mrFlags"][:k] & 0xFFFF) <= 16).all()
        small = (items["mrFlags"] & BF_ITEM_SMALL) != 0
        if small.any():      # small items are the tail of the list (they get their own launch, four to a wavefront)
            first = int(np.argmax(small))
            assert small[first:].all() and not transpose and info.dtype != 0
            assert ((items["mrFlags"][first:] & 0xFFFF) <= 2 * epl).all() and (items["numPieces"][first:] <= 16).all()
        for it in items:
            mr = int(it["mrFlags"]) & 0xFFFF
            mr_pad = mr if transpose else (mr + epl - 1) // epl * epl
            acc = np.zeros((mr, nrhs), dtype=dt)
            if int(it["mrFlags"]) & BF_ITEM_SMALL:
                assert int(it["mrFlags"]) & BF_ITEM_ROWMAJOR and not int(it["mrFlags"]) & BF_ITEM_MERGED      # small items: row-major pieces
            if int(it["mrFlags"]) & BF_ITEM_MERGED and (int(it["mrFlags"]) & BF_ITEM_MERGED or any(not int(pc["flags"]) & BF_PIECE_IDENTITY for pc in pieces[int(it["pieceBegin"]):int(it["pieceBegin"]) + int(it["numPieces"])])):
                # the kernel reads the dense pieces as ONE mr_pad x n block starting at the first one's offset
                mine = pieces[int(it["pieceBegin"]):int(it["pieceBegin"]) + int(it["numPieces"])]
                dense = [pc for pc in mine if not int(pc["flags"]) & BF_PIECE_IDENTITY]
                assert not transpose and info.dtype != 0 and len(mine) <= 64 and dense
                assert sum(int(pc["ncols"]) for pc in dense) <= (384 if int(it["mrFlags"]) & BF_ITEM_SMALL else 256)
                nxt = int(dense[0]["dataOff"])
                for pc in dense:
                    assert int(pc["dataOff"]) == nxt and not int(pc["flags"]) & BF_PIECE_ROWMAJOR
                    nxt += mr_pad * int(pc["ncols"])
            for pc in pieces[int(it["pieceBegin"]):int(it["pieceBegin"]) + int(it["numPieces"])]:
                src = x if (int(pc["flags"]) & BF_PIECE_IN_X) else temp
                io, n = int(pc["inOff"]), int(pc["ncols"])
                if int(pc["flags"]) & BF_PIECE_IDENTITY:
                    acc += src[io:io + mr]
                    continue
                d0 = int(pc["dataOff"])
                rowmajor = bool(int(pc["flags"]) & BF_PIECE_ROWMAJOR)
                if rowmajor:
                    # few-row leaves of real operands: element (r, c) = arena[d0 + r * ld + c], rows padded to the lane granule
                    ld = int(pc["ld"])
                    assert info.dtype != 0 and ld % epl == 0 and d0 % epl == 0
                    if transpose:       # n rows of the forward piece, mr of its columns starting at d0
                        assert d0 + (n - 1) * ld + (mr + epl - 1) // epl * epl <= len(arena)
                        idx = d0 + np.arange(n)[None, :] * ld + np.arange(mr)[:, None]
                        acc += arena[idx] @ src[io:io + n]
                    else:
                        assert int(it["mrFlags"]) & BF_ITEM_ROWMAJOR and ld >= n and mr <= 2 * epl
                        blk = arena[d0:d0 + mr * ld].reshape(mr, ld)
                        assert not blk[:, n:].any(), "row padding must be zero"
                        acc += blk[:, :n] @ src[io:io + n]
                    continue
                if transpose:
                    # lanes j < mr on the columns of a forward piece: element (step s, lane j) = arena[d0 + j*ld + s]
                    ld = int(pc["ld"])
                    assert ld >= n and mr <= 128
                    # what bfStageKernelT may touch: columns j < mr, units clamped into the column
                    assert d0 + (mr - 1) * ld + (n + epl - 1) // epl * epl <= len(arena), "transposed piece reaches past the leaf arena"
                    idx = d0 + np.arange(mr)[:, None] * ld + np.arange(n)[None, :]
                    acc += arena[idx] @ src[io:io + n]
                    continue
                assert n <= info.xcap
                a = arena[d0:d0 + mr_pad * n].reshape(n, mr_pad).T[:mr]
                acc += a @ src[io:io + n]
            dst = y if (int(it["mrFlags"]) & BF_ITEM_OUT_Y) else temp
            oo = int(it["outOff"])
            dst[oo:oo + mr] = acc
        for r in range(int(sv.numReduce)):
            rv = _capi.BfhipReduceView()
            rv.structSize = C.sizeof(rv)
            _capi.check(lib.bfhipPlanGetReduce(op.handle, s, r, C.byref(rv)))
            row_iv = _view(rv.rowInterval, int(rv.numRows), np.dtype("<u4"))
            iv_begin = _view(rv.ivBegin, int(rv.numIntervals) + 1, np.dtype("<u4"))
            bias = _view(rv.srcBias, int(rv.numSrc), np.dtype("<i8"))
            dest = y if rv.destIsY else temp[int(rv.destOff):]
            rows = np.arange(int(rv.numRows))
            out = np.zeros((int(rv.numRows), nrhs), dtype=dt)
            keep = row_iv != 0xFFFFFFFF           # BF_REDUCE_SKIP: written directly by the one group that owns the row
            assert keep.all() or info.dtype != 0
            riv = np.where(keep, row_iv, 0)
            cnt = np.where(keep, iv_begin[riv + 1] - iv_begin[riv], 0)
            for k in range(int(cnt.max()) if len(cnt) else 0):
                sel = cnt > k
                src_rows = bias[iv_begin[riv[sel]] + k] + rows[sel]
                out[sel] += temp[src_rows]
            dest[:int(rv.numRows)][keep] = out[keep]
    return y[:, 0] if one_d else y
