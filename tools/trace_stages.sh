#!/bin/bash
# Per-dispatch timeline of one apply (kernel durations and the gaps between them) from rocprofv3 --kernel-trace.
# This is where DESIGN.md's per-stage numbers of the transposed plan and the launch-gap measurements come from.
#   usage (on the GPU box): tools/trace_stages.sh forward|transposed <bench.py arguments>
#   e.g.  tools/trace_stages.sh transposed --workload streamer --npoints 262144 --lmax 127
#         tools/trace_stages.sh forward --npoints 65536
# Runs bench.py with 3 steps (every apply bracketed by events: expect ~6 us per event between kernels) and prints the
# last apply of the chosen direction.
set -u
DIR=${1:-forward}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/trace_$DIR
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
EXTRA=""; [ "$DIR" = "transposed" ] && EXTRA="--adjoint"
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 $R/bench.py "$@" $EXTRA --steps 3 --warmup 1 --no-cpu-baseline --no-extra > $OUT/bench.json 2> $OUT/bench.log
python3 - "$OUT" "$DIR" <<'PY'
import csv, glob, sys
out, direction = sys.argv[1], sys.argv[2]
f = sorted(glob.glob(out + "/trace/**/*kernel_trace.csv", recursive=True))[-1]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
ours = [r for r in rows if "bfStageKernel" in r["Kernel_Name"] or "bfReduceKernel" in r["Kernel_Name"]]
want_t = direction == "transposed"
is_t = lambda r: "bfStageKernelT" in r["Kernel_Name"]
# the last stage kernel of the chosen direction, then back while dispatches are < 200 us apart and of that direction
idx = max(i for i, r in enumerate(ours) if ("bfStageKernel" in r["Kernel_Name"]) and is_t(r) == want_t)
if idx + 1 < len(ours) and "bfReduceKernel" in ours[idx + 1]["Kernel_Name"]:
    idx += 1
sel = [idx]
for i in range(idx - 1, -1, -1):
    r = ours[i]
    if "bfStageKernel" in r["Kernel_Name"] and is_t(r) != want_t: break
    if int(ours[sel[-1]]["Start_Timestamp"]) - int(r["End_Timestamp"]) > 60000: break
    sel.append(i)
sel.reverse()
prev = None; tot = gaps = 0.0
for i in sel:
    r = ours[i]; s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev) / 1e3 if prev is not None else 0.0
    print("%-56s %10.2f us   gap %7.2f us   grid=%s" % (r["Kernel_Name"][:56], (e - s) / 1e3, gap, r.get("Grid_Size_X", r.get("Grid_Size", "?"))))
    tot += (e - s) / 1e3; gaps += gap; prev = e
print("dispatches %d   sum of durations %.1f us   sum of gaps %.1f us" % (len(sel), tot, gaps))
PY
