#!/bin/bash
# A/B of library variants (butterfly_amd/csrc/exp/libbfhip_<v>.so; "base" = the product library) on the streamed operand.
#   usage (GPU box): bash tools/ab_variants_streamer.sh <tag> <variant> [<variant> ...]
# (a variant is built HERE, before gpurun snapshots the tree: make -C butterfly_amd/csrc variant V=<name> DEFS="-D..." -> exp/libbfhip_<name>.so;
#  the switches are listed next to that target in butterfly_amd/csrc/Makefile)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$1; shift; mkdir -p $O; cd $R
for v in "$@"; do
  if [ $v = base ]; then unset BFHIP_LIB_PATH; else export BFHIP_LIB_PATH=$R/butterfly_amd/csrc/exp/libbfhip_$v.so; fi
  timeout -k 10 400 python bench.py --workload streamer --adjoint --steps 10 --warmup 2 --no-cpu-baseline --no-extra > $O/st32_$v.out 2> $O/st32_$v.err || exit 1
  timeout -k 10 400 python bench.py --workload streamer --dtype f64 --adjoint --steps 10 --warmup 2 --no-cpu-baseline --no-extra > $O/st64_$v.out 2> $O/st64_$v.err || exit 1
  python - <<PY
import json
a = json.load(open("$O/st32_$v.out")); b = json.load(open("$O/st64_$v.out"))
print("$v f32 fwd", round(a["ms_per_step"], 3), round(a["roofline"]["frac"], 4), "adj", round(a["adjoint"]["ms_per_apply"], 3), round(a["adjoint"]["frac_of_hbm_peak"], 4),
      "| f64 fwd", round(b["ms_per_step"], 3), round(b["roofline"]["frac"], 4), "adj", round(b["adjoint"]["ms_per_apply"], 3), round(b["adjoint"]["frac_of_hbm_peak"], 4))
PY
done
