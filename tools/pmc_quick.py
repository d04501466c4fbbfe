#!/usr/bin/env python3
"""Per-kernel sums of the counters of rocprofv3 --pmc passes (csv), next to the kernel-trace durations of the same pass:
HBM fetch bytes per launch (FETCH_SIZE KiB x 2 on gfx950, MI355X_MICROARCH.md), core clock = GRBM_GUI_ACTIVE / duration,
matrix-pipe busy fraction = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs).  usage: pmc_quick.py <pass dir> ..."""
import collections, csv, glob, os, sys
WANT = os.environ.get("PMC_QUICK_FILTER", "Stage")      # substring of the kernel names to report
for d in sys.argv[1:]:
    cc = sorted(glob.glob(d + "/*/*_counter_collection.csv"))
    kt = sorted(glob.glob(d + "/*/*_kernel_trace.csv"))
    if not cc or not kt:
        print(d, "no csv"); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(int)
    for r in csv.DictReader(open(cc[-1])):
        k = r["Kernel_Name"].split("(")[0]; agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    dur = collections.defaultdict(float)
    for r in csv.DictReader(open(kt[-1])):
        k = r["Kernel_Name"].split("(")[0]; dur[k] += (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) * 1e-9; n[k] += 1
    if os.environ.get("PMC_QUICK_PER_DISPATCH") == "1":      # one line per dispatch (a probe binary launches one kernel name for several cases)
        per = collections.defaultdict(lambda: collections.defaultdict(float)); nm = {}
        for r in csv.DictReader(open(cc[-1])):
            per[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"]); nm[r["Dispatch_Id"]] = r["Kernel_Name"].split("(")[0]
        du = {r["Dispatch_Id"]: (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) * 1e-9 for r in csv.DictReader(open(kt[-1]))}
        for did in sorted(per, key=int):
            c = per[did]
            if WANT not in nm[did] or did not in du: continue
            out = [d.split("/")[-1], "dispatch", did, nm[did][:40], "ms %.3f" % (du[did] * 1e3)]
            if "GRBM_GUI_ACTIVE" in c: out += ["clock GHz %.3f" % (c["GRBM_GUI_ACTIVE"] / 8 / du[did] / 1e9)]
            if "SQ_VALU_MFMA_BUSY_CYCLES" in c and "GRBM_GUI_ACTIVE" in c: out += ["mfma busy %.3f" % (c["SQ_VALU_MFMA_BUSY_CYCLES"] / (c["GRBM_GUI_ACTIVE"] / 8 * 1024))]
            if "SQ_VALU_MFMA_BUSY_CYCLES" in c: out += ["busy GHz %.3f" % (c["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / du[did] / 1e9)]
            print(*out)
        continue
    for k, c in agg.items():
        if WANT not in k: continue
        out = [d.split("/")[-1], k, "launches", n[k], "ms/launch %.3f" % (dur[k] / max(n[k], 1) * 1e3)]
        if "FETCH_SIZE" in c: out += ["fetch GB/launch %.3f" % (c["FETCH_SIZE"] * 1024 * 2 / max(n[k], 1) / 1e9)]
        if "GRBM_GUI_ACTIVE" in c: out += ["clock GHz %.3f" % (c["GRBM_GUI_ACTIVE"] / 8 / dur[k] / 1e9)]
        if "SQ_VALU_MFMA_BUSY_CYCLES" in c and "GRBM_GUI_ACTIVE" in c: out += ["mfma busy %.3f" % (c["SQ_VALU_MFMA_BUSY_CYCLES"] / (c["GRBM_GUI_ACTIVE"] / 8 * 1024))]
        print(*out)
