#!/usr/bin/env python3
"""What the items of a 64-RHS plan share (CPU only: a plan-only operator and its stage views).

Per stage of the fac_helm2 benchmark operand compiled for blocks of right-hand sides:
  * the bundle table (bfPlanBundles): how much of the work sits in shared bundles (four neighbours with equal inputs);
  * X panel reads (32-row passes x columns x 1 KiB at 64 RHS) against the leaf bytes;
  * what STACKING the items of equal inputs into common 16-row slabs would buy (part 3): the row padding of the
    MFMA tiles and the X panel reads, now and stacked.
usage: python tools/plan_x_sharing.py [N]          (N = 65536 takes ~1 minute, 262144 ~5)"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from butterfly_amd import _capi, helm2_structure as hs          # noqa: E402
from butterfly_amd.operator import HipOperator                  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
desc, perm = hs.native_multilevel_structure(hs.circle_points(n), n // 16)
op = HipOperator.from_desc(desc, None, root=desc.root, flags=_capi.FLAG_PLAN_ONLY, max_rhs=64, seed=1)
lib = _capi.load()
info = _capi.BfhipPlanInfo(); info.structSize = C.sizeof(info)
_capi.check(lib.bfhipPlanGetInfo(op.handle, C.byref(info)))
tot = dict(slab=0, real=0, stslab=0, xr=0, stxr=0)
for s in range(int(info.numStages)):
    sv = _capi.BfhipStageView(); sv.structSize = C.sizeof(sv)
    _capi.check(lib.bfhipPlanGetStage(op.handle, s, C.byref(sv)))
    items = np.frombuffer((C.c_char * (int(sv.numItems) * 16)).from_address(sv.items), dtype=_capi.ITEM_DTYPE)
    pieces = np.frombuffer((C.c_char * (int(sv.numPieces) * 24)).from_address(sv.pieces), dtype=_capi.PIECE_DTYPE)
    bb = np.frombuffer((C.c_char * ((int(sv.numBundles) + 1) * 4)).from_address(sv.bundleBegin), dtype=np.uint32)
    mixed = (bb[:-1] >> 31) != 0
    cnt = np.diff((bb & 0x7FFFFFFF).astype(np.int64))
    mr = (items["mrFlags"] & 0xFFFF).astype(np.int64)
    pb, npc = items["pieceBegin"].astype(np.int64), items["numPieces"].astype(np.int64)
    cs = np.concatenate([[0], np.cumsum(pieces["ncols"].astype(np.int64))])
    cols = cs[pb + npc] - cs[pb]
    work = mr * cols
    shared = work[np.repeat(~mixed, cnt)].sum() / max(work.sum(), 1)
    groups = {}
    for i in range(len(items)):
        if npc[i]:
            key = (pieces["inOff"][pb[i]:pb[i] + npc[i]].tobytes(), pieces["ncols"][pb[i]:pb[i] + npc[i]].tobytes(), (pieces["flags"][pb[i]:pb[i] + npc[i]] & 3).tobytes())
            groups.setdefault(key, []).append(i)
    slab = real = stslab = xr = stxr = 0
    for idx in groups.values():
        c, rows = int(cols[idx[0]]), int(mr[idx].sum())
        slab += int(((mr[idx] + 15) // 16).sum()) * 16 * c; real += rows * c; stslab += (rows + 15) // 16 * 16 * c
        xr += int(((mr[idx] + 31) // 32).sum()) * c; stxr += (rows + 31) // 32 * c
    for k_, v in (("slab", slab), ("real", real), ("stslab", stslab), ("xr", xr), ("stxr", stxr)): tot[k_] += v
    print(f"stage {s}: {len(items)} items, {len(groups)} distinct inputs, {len(cnt)} bundles ({int((~mixed).sum())} shared = {shared:.3f} of the work); "
          f"leaves {real * 16 / 1e9:.2f} GB, X panel reads {xr * 1024 / 1e9:.2f} GB (stacked {stxr * 1024 / 1e9:.2f}); row padding {slab / max(real, 1):.3f} (stacked {stslab / max(real, 1):.3f})", flush=True)
print(f"apply: leaves {tot['real'] * 16 / 1e9:.1f} GB, X panel reads {tot['xr'] * 1024 / 1e9:.1f} GB (stacked {tot['stxr'] * 1024 / 1e9:.1f}); row padding {tot['slab'] / tot['real']:.4f} (stacked {tot['stslab'] / tot['real']:.4f})")
op.close()
