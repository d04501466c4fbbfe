"""Where inside a launch the time goes: runs one apply with BFHIP_TIMELINE_FILE set (bfhip_device.hip: every complex128
stage launch records when each item started and ended, 100 MHz ticks) and prints, per launch, the bytes, the span, the
rate, how long the launch took to reach half / 90 % of its peak concurrency, how long the tail below 50 % concurrency
lasted, the share of the items' lifetime spent before / between / after streaming, and (--bins) the bytes moved in 10 us bins.
  python tools/timeline.py --npoints 65536 [--emulate-world 8 --emulate-rank 3]"""
import argparse, json, os, sys, tempfile
import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
# the executors this tool drives live in the experimental build only (make -C butterfly_amd/csrc experimental)
os.environ.setdefault("BFHIP_LIB_PATH", os.path.join(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."), "butterfly_amd", "csrc", "libbfhip_exp.so"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--npoints", type=int, default=65536)
    ap.add_argument("--world", type=int, default=1)
    ap.add_argument("--rank", type=int, default=0)
    ap.add_argument("--bins", action="store_true")
    ap.add_argument("--raw", default=None, help="save the raw per-item arrays (npz) here")
    args = ap.parse_args()
    import torch
    from butterfly_amd import helm2_structure as hs, dist
    from butterfly_amd.operator import HipOperator
    n, k = args.npoints, args.npoints / 16.0
    desc, _ = hs.native_multilevel_structure(hs.circle_points(n), k)
    kw = {}
    if args.world > 1:
        cuts, _ = dist.row_partition(desc, args.world)
        kw["row_range"] = (cuts[args.rank], cuts[args.rank + 1])
    op = HipOperator.from_desc(desc, None, seed=1, **kw)
    x = torch.randn(n, dtype=torch.complex128, device="cuda")
    for _ in range(3):
        op.apply_device(x)
    torch.cuda.synchronize()
    path = tempfile.mktemp(suffix=".timeline")
    os.environ["BFHIP_TIMELINE_FILE"] = path
    op.apply_device(x)
    torch.cuda.synchronize()
    del os.environ["BFHIP_TIMELINE_FILE"]
    lines = open(path).read().split("\n")
    os.remove(path)
    out, i, raws = [], 0, {}
    while i < len(lines):
        if not lines[i].startswith("launch"):
            i += 1
            continue
        ni = int(lines[i].split()[1])
        a = np.array([[int(v) for v in lines[i + 1 + j].split()] for j in range(ni)], dtype=np.int64).reshape(ni, 9)
        i += 1 + ni
        raws[f"launch{len(out)}"] = a
        t0, t1 = a[:, 0] * 10e-3, a[:, 1] * 10e-3                    # us
        nb = a[:, 2] * a[:, 3] * 16.0
        base = t0.min()
        t0, t1 = t0 - base, t1 - base
        span = t1.max()
        ev = np.concatenate([np.stack([t0, np.ones(ni)], 1), np.stack([t1, -np.ones(ni)], 1)])
        ev = ev[np.argsort(ev[:, 0], kind="stable")]
        conc = np.cumsum(ev[:, 1])
        peak = conc.max()
        t_half = ev[np.argmax(conc >= 0.5 * peak), 0]
        t_90 = ev[np.argmax(conc >= 0.9 * peak), 0]
        last_above_half = ev[len(conc) - 1 - np.argmax(conc[::-1] >= 0.5 * peak), 0]
        rec = {"items": ni, "mb": nb.sum() / 1e6, "span_us": span, "gbs": nb.sum() / span / 1e3, "peak_waves": int(peak),
               "us_to_half_peak": t_half, "us_to_90pct_peak": t_90, "tail_below_half_us": span - last_above_half,
               "item_us_median": float(np.median(t1 - t0)), "item_us_max": float((t1 - t0).max()), "item_kb_median": float(np.median(nb)) / 1e3,
               "item_kb_max": float(nb.max()) / 1e3, "first_item_end_us": float(t1.min())}
        # inside the items (100 MHz stamps; items with a dense piece): record + first descriptor, first x gather, streaming, reduce + store
        ok = (a[:, 5] > 0) & (a[:, 6] > 0) & (a[:, 8] > 0)
        if ok.any():
            td, tx, te = a[ok, 5] * 10e-3 - base, a[ok, 6] * 10e-3 - base, a[ok, 8] * 10e-3 - base
            life = (t1 - t0)[ok].sum()
            rec["share_of_item_time"] = {"record_and_descriptor": float((td - t0[ok]).sum() / life), "first_x_gather": float((tx - td).sum() / life),
                                         "pieces": float((te - tx).sum() / life), "reduce_and_store": float((t1[ok] - te).sum() / life)}
        if args.bins:
            # bytes attributed uniformly over each item's lifetime, 10 us bins
            edges = np.arange(0, span + 10, 10.0)
            acc = np.zeros(len(edges) - 1)
            for s, e, b in zip(t0, t1, nb):
                lo, hi = np.searchsorted(edges, s, "right") - 1, np.searchsorted(edges, e, "left")
                for q in range(max(lo, 0), min(hi, len(acc))):
                    ov = min(e, edges[q + 1]) - max(s, edges[q])
                    if ov > 0:
                        acc[q] += b * ov / max(e - s, 1e-9)
            rec["gbs_per_10us_bin"] = [round(v / 10 / 1e3, 0) for v in acc]
        out.append(rec)
    if args.raw:
        np.savez_compressed(args.raw, **raws)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
