#!/bin/bash
# usage: ab.sh tag variants...
# (a variant is built HERE, before gpurun snapshots the tree: make -C butterfly_amd/csrc variant V=<name> DEFS="-D..." -> exp/libbfhip_<name>.so;
#  the switches are listed next to that target in butterfly_amd/csrc/Makefile)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$1; shift; mkdir -p $O; cd $R
for v in "$@"; do
  if [ $v = base ]; then unset BFHIP_LIB_PATH; else export BFHIP_LIB_PATH=$R/butterfly_amd/csrc/exp/libbfhip_$v.so; fi
  timeout -k 10 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-extra > $O/h_$v.out 2> $O/h_$v.err || exit 1
  timeout -k 10 300 python bench.py --npoints 65536 --steps 100 --warmup 5 --no-cpu-baseline --no-extra > $O/n_$v.out 2> $O/n_$v.err || exit 1
  timeout -k 10 600 python bench.py --emulate-world 8 --emulate-rank -1 --steps 20 --no-extra --no-cpu-baseline --shard rowsum > $O/s_$v.out 2> $O/s_$v.err || exit 1
  python - <<PY
import json
h = json.load(open("$O/h_$v.out")); n = json.load(open("$O/n_$v.out")); s = json.load(open("$O/s_$v.out"))
print("$v headline", round(h["ms_per_step"], 4), round(h["roofline"]["frac"], 4), "| n65536", round(n["ms_per_step"], 4), round(n["roofline"]["frac"], 4), "| shards8 slowest", round(s["emulated_shard"]["slowest_ms"], 4))
PY
done
