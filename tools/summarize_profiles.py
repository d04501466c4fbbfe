#!/usr/bin/env python3
"""Condense gpurun_out/prof_<tag>/ (tools/profile_round.sh) into the small files committed under profiles/."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r1"
src = f"gpurun_out/prof_{tag}"
os.makedirs("profiles", exist_ok=True)


def newest(pattern):
    """A pass directory may hold an earlier run of the same pass (files are named by pid): only the newest run counts."""
    files = sorted(glob.glob(pattern), key=os.path.getmtime)
    return files[-1:]


def counters(name):
    files = newest(f"{src}/{name}/*/*_counter_collection.csv")
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    if not files:
        return {}
    for r in csv.DictReader(open(files[0])):
        kn = r["Kernel_Name"].split("(")[0]
        agg[kn][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: {"dispatches": len(v), "sum": sum(v), "mean": sum(v) / len(v)} for c, v in d.items()} for k, d in agg.items()}


out = {"tag": tag, "note": "FETCH_SIZE/WRITE_SIZE in KiB; on gfx950 FETCH_SIZE counts half the bytes of a 16-B/lane streaming read "
       "(MI355X_MICROARCH.md, HBM): fetch bytes = FETCH_SIZE*1024*2; WRITE_SIZE exact.  SQ_* are summed over SEs/XCDs as rocprofv3 reports them."}
for name in ("stats_r1", "stats_r64", "stats_st", "stats_n65536", "stats_adj", "stats_stT"):
    for f in newest(f"{src}/{name}/*/*_kernel_stats.csv"):
        shutil.copy(f, f"profiles/{tag}_{name}_kernel_stats.csv")
    if os.path.exists(f"{src}/{name}.json"):
        shutil.copy(f"{src}/{name}.json", f"profiles/{tag}_{name}_bench_under_rocprof.json")
for name in ("fetch_r1", "write_r1", "lds_r1", "mfma_r64", "fetch_r64", "write_r64", "fetch_st", "write_st", "fetch_n65536", "write_n65536",
             "fetch_adj", "write_adj", "fetch_stT", "write_stT", "sq_st", "sq_stT"):
    out[name] = counters(name)


def kernel_seconds(name, prefix):
    """Sum of (end - start) over the dispatches of kernels whose name starts with `prefix`, from the kernel trace of pass `name`."""
    tot, cnt = 0.0, 0
    for f in newest(f"{src}/{name}/*/*_kernel_trace.csv"):
        for r in csv.DictReader(open(f)):
            if r["Kernel_Name"].split("(")[0].startswith(prefix):
                tot += (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) * 1e-9
                cnt += 1
    return tot, cnt


def family_per_apply(fetch, write, prefix, applies_json):
    """All kernels whose name starts with `prefix` (a transposed apply is several kernels per stage), per apply."""
    try:
        f = sum(v["FETCH_SIZE"]["sum"] for k, v in out[fetch].items() if k.startswith(prefix)) * 1024 * 2
        w = sum(v["WRITE_SIZE"]["sum"] for k, v in out[write].items() if k.startswith(prefix)) * 1024
        nf = sum(v["FETCH_SIZE"]["dispatches"] for k, v in out[fetch].items() if k.startswith(prefix))
        b = json.load(open(f"{src}/{applies_json}.json"))
        # bench.py --adjoint: 2 warm-up transposed applies + `steps` timed ones; on the streamed operand also cov_matvec
        # (2 warm-up + `steps` timed products + 1 checked against the separate applies: one transposed apply each)
        applies = b["steps"] + 2 + ((b["steps"] + 3) if "cov_matvec" in b else 0)
        if not nf:
            return None
        return {"fetch_bytes_corrected": f / applies, "write_bytes": w / applies, "hbm_bytes": (f + w) / applies, "dispatches": nf, "applies": applies,
                "algorithmic_bytes": b["config"]["leaf_bytes"], "ratio": (f + w) / applies / b["config"]["leaf_bytes"]}
    except (KeyError, FileNotFoundError):
        return None


def per_launch(fetch, write, kern):
    try:
        f = out[fetch][kern]["FETCH_SIZE"]["mean"] * 1024 * 2
        w = out[write][kern]["WRITE_SIZE"]["mean"] * 1024
        return {"fetch_bytes_corrected": f, "write_bytes": w, "hbm_bytes": f + w}
    except KeyError:
        return None


out["bfStageKernelC128_per_launch"] = per_launch("fetch_r1", "write_r1", "bfStageKernelC128")
out["bfStageKernelC128Mfma_per_launch"] = per_launch("fetch_r64", "write_r64", "bfStageKernelC128Mfma")
out["bfStageKernelC128_n65536_per_launch"] = per_launch("fetch_n65536", "write_n65536", "bfStageKernelC128")
out["bfStageKernelT_c128_adjoint_per_apply"] = family_per_apply("fetch_adj", "write_adj", "void bfStageKernelT", "fetch_adj")
out["bfStageKernelT_f32_streamer_adjoint_per_apply"] = family_per_apply("fetch_stT", "write_stT", "void bfStageKernelT", "fetch_stT")


def family_per_launch(fetch, write, prefixes):
    """Mean HBM bytes per launch over a family of kernels (a real stage is ONE launch of bfStageKernelRealBoth, or of
    bfStageKernelReal / bfStageKernelSmall alone when it has one kind of item only)."""
    try:
        sel = lambda d, c: [v[c] for k, v in d.items() if any(k.startswith(q) for q in prefixes) and c in v]
        ff, ww = sel(out[fetch], "FETCH_SIZE"), sel(out[write], "WRITE_SIZE")
        n = sum(v["dispatches"] for v in ff)
        if not n or n != sum(v["dispatches"] for v in ww):
            return None
        f, w = sum(v["sum"] for v in ff) * 1024 * 2 / n, sum(v["sum"] for v in ww) * 1024 / n
        return {"fetch_bytes_corrected": f, "write_bytes": w, "hbm_bytes": f + w, "dispatches": n}
    except KeyError:
        return None


out["bfStageKernelReal_f32_streamer_per_launch"] = family_per_launch("fetch_st", "write_st", ("void bfStageKernelReal", "void bfStageKernelSmall"))
for nm, key in (("stats_r1", "bfStageKernelC128_per_launch"), ("stats_r64", "bfStageKernelC128Mfma_per_launch"), ("stats_st", "bfStageKernelReal_f32_streamer_per_launch"),
                ("stats_n65536", "bfStageKernelC128_n65536_per_launch")):
    p = f"{src}/{nm}.json"
    if os.path.exists(p) and out.get(key):
        b = json.load(open(p))
        rl = b["roofline"]
        per_apply = rl.get("algorithmic_bytes_per_apply")
        if per_apply is None and "hbm_gbs_algorithmic" in rl:       # RHS-block lines carry the rate: bytes = rate x kernel time
            per_apply = rl["hbm_gbs_algorithmic"] * 1e9 * rl["kernel_ms_per_apply"] / 1e3
        if per_apply is not None:
            out[key]["algorithmic_bytes"] = per_apply / rl["launches_per_apply"]
            out[key]["ratio"] = out[key]["hbm_bytes"] / out[key]["algorithmic_bytes"]


def wave_cycles(name, prefix):
    """Where the wavefronts of the kernels starting with `prefix` spent their cycles (SQ counters, quad-cycles summed over the chip):
    parked on s_waitcnt (memory), stalled at issue, issuing; vector-memory and VALU instructions per 1 KB streamed."""
    ks = {k: v for k, v in out.get(name, {}).items() if k.startswith(prefix) and "SQ_WAVE_CYCLES" in v}
    if not ks:
        return None
    tot = lambda c: sum(v[c]["sum"] for v in ks.values())
    wc = tot("SQ_WAVE_CYCLES")
    secs, cnt = kernel_seconds(name, prefix)
    r = {"kernels": sorted(ks), "dispatches": cnt, "wait_any_frac": tot("SQ_WAIT_ANY") / wc, "wait_inst_any_frac": tot("SQ_WAIT_INST_ANY") / wc,
         "active_inst_any_frac": tot("SQ_ACTIVE_INST_ANY") / wc, "insts_vmem_rd": tot("SQ_INSTS_VMEM_RD"), "insts_valu": tot("SQ_INSTS_VALU"),
         "valu_per_vmem_rd": tot("SQ_INSTS_VALU") / max(tot("SQ_INSTS_VMEM_RD"), 1)}
    if secs > 0:
        # mean resident wavefronts per SIMD = wave quad-cycles * 4 / (active cycles per XCD * 1024 SIMDs)
        gui = tot("GRBM_GUI_ACTIVE") / 8
        r["sustained_clock_ghz"] = gui / secs * 1e-9
        r["mean_waves_per_simd"] = wc * 4 / (gui * 1024)
    return r


out["bfStageKernelReal_f32_streamer_wave_cycles"] = wave_cycles("sq_stT", "void bfStageKernelReal") or wave_cycles("sq_st", "void bfStageKernelReal")
out["bfStageKernelT_f32_streamer_wave_cycles"] = wave_cycles("sq_stT", "void bfStageKernelT")
try:
    m = out["mfma_r64"]["bfStageKernelC128Mfma"]
    busy, gui = m["SQ_VALU_MFMA_BUSY_CYCLES"]["sum"], m["GRBM_GUI_ACTIVE"]["sum"]
    out["mfma_util_percent"] = {"formula": "SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE(max over XCDs ~ sum/8) * 1024 SIMDs) * 100",
                                "value": busy / (gui / 8 * 1024) * 100,
                                "mfma_f64_ops_x512_flops": m["SQ_INSTS_VALU_MFMA_MOPS_F64"]["sum"] * 512}
    # sustained core clock under the matrix-core kernel: cycles the GPU was active (per XCD) / the time its dispatches took
    secs, cnt = kernel_seconds("mfma_r64", "bfStageKernelC128Mfma")
    if secs > 0:
        out["mfma_sustained_clock"] = {"formula": "GRBM_GUI_ACTIVE (sum over 8 XCDs / 8) / sum of the dispatches' durations (kernel trace of the same pass)",
                                       "ghz": gui / 8 / secs * 1e-9, "dispatches": cnt, "seconds": secs, "peak_ghz": 2.4,
                                       "mfma_busy_fraction": busy / (gui / 8 * 1024)}
except KeyError:
    pass
json.dump(out, open(f"profiles/{tag}_pmc_summary.json", "w"), indent=1)
print(json.dumps({k: out[k] for k in out if k.endswith("per_launch") or k.endswith("per_apply") or k.endswith("wave_cycles") or k.startswith("mfma_")}, indent=1))
