#!/usr/bin/env python3
"""Experiment (round 3): does running independent sub-operators on separate HIP streams hide the ramp / tail of the
short stage kernels of small operands?  The top-level (row, col) blocks of a fac_helm2 operand are independent up to
the final sum into y, so L "lanes" (LPT over the blocks) are L operators whose stage s kernels can overlap each other's
tails.  Times one stream vs L streams on the same leaf bytes."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=65536)
    ap.add_argument("--lanes", type=int, nargs="+", default=[2, 3, 4])
    ap.add_argument("--world", type=int, default=1, help="take rank 0's blocks-mode shard of a WORLD-rank job as the operand")
    ap.add_argument("--steps", type=int, default=50)
    args = ap.parse_args()
    import numpy as np
    import torch
    from butterfly_amd import helm2_structure as hs
    from butterfly_amd.dist import assign_row_blocks, block_weights
    from butterfly_amd.operator import HipOperator
    n = args.n
    desc, perm = hs.native_multilevel_structure(hs.circle_points(n), n / 16.0)
    bw = block_weights(desc)
    blocks = list(range(len(bw)))
    if args.world > 1:
        owner, _ = assign_row_blocks(bw, args.world)
        blocks = [i for i in blocks if owner[i] == 0]
    rng = np.random.default_rng(0)
    x = torch.from_numpy((rng.standard_normal(n) + 1j * rng.standard_normal(n)) / np.sqrt(2)).cuda()

    def timed(fn):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / args.steps * 1e3

    out = {"n": n, "world": args.world, "blocks": len(blocks)}
    root = hs.shard_desc_blocks(desc, blocks)
    op = HipOperator.from_desc(desc, None, root=root, seed=1)
    y = torch.empty(n, dtype=torch.complex128, device="cuda")
    gb = op.stats()["leafBytes"] / 1e9
    ms = timed(lambda: op.apply_device(x, y))
    out["one_stream"] = {"ms": ms, "gbs": gb / ms * 1e3, "leaf_gb": gb}
    y_ref = y.clone()
    op.close()
    main_stream = torch.cuda.current_stream()
    for L in args.lanes:
        w = [bw[i] for i in blocks]
        owner, loads = assign_row_blocks(w, L)
        ops, ys, streams = [], [], []
        for l in range(L):
            r = hs.shard_desc_blocks(desc, [blocks[i] for i in range(len(blocks)) if owner[i] == l])
            ops.append(HipOperator.from_desc(desc, None, root=r, seed=1))
            ys.append(torch.empty(n, dtype=torch.complex128, device="cuda"))
            streams.append(torch.cuda.Stream())
        ev0 = torch.cuda.Event()
        evs = [torch.cuda.Event() for _ in range(L)]

        def step():
            ev0.record(main_stream)
            for l in range(L):
                streams[l].wait_event(ev0)
                ops[l].apply_device(x, ys[l], stream=streams[l])
                evs[l].record(streams[l])
            for l in range(L):
                main_stream.wait_event(evs[l])
            torch.add(ys[0], ys[1], out=y)
            for l in range(2, L):
                y.add_(ys[l])
        ms = timed(step)
        err = float(torch.linalg.norm(y - y_ref) / torch.linalg.norm(y_ref))
        out[f"lanes{L}"] = {"ms": ms, "gbs": gb / ms * 1e3, "rel_vs_one_stream": err, "lane_gb": [l * 16 / 1e9 for l in loads]}
        for o in ops:
            o.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
