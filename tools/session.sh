#!/bin/bash
# One gpurun call = a list of bounded steps; a step that is killed at its limit ends the call (nothing further touches the GPU).
#   gpurun --timeout 1200 -- 'bash tools/session.sh <tag> <step> [<step> ...]'     steps are functions below
set -u
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
step() { # name limit cmd...
  local name=$1 lim=$2; shift 2
  echo "== $name"
  timeout -k 10 $lim "$@" > $O/$name.out 2> $O/$name.err
  local rc=$?
  echo "$name exit=$rc"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step $name hit its limit: stopping"; tail -5 $O/$name.err; exit 1; fi
}
B="python bench.py"
gputests()   { step pytest 900 python -m pytest tests -m gpu -x -q; tail -5 $O/pytest.out; }
n65536()     { step bench_n65536 300 $B --npoints 65536 --steps 50 --warmup 5 --no-cpu-baseline --no-extra; grep "stage " $O/bench_n65536.err; cat $O/bench_n65536.out | cut -c1-400; }
shards8()    { step shards8_rows 600 $B --emulate-world 8 --emulate-rank -1 --steps 20 --no-extra --no-cpu-baseline --shard rows;
               step shards8_blocks 600 $B --emulate-world 8 --emulate-rank -1 --steps 20 --no-extra --no-cpu-baseline --shard blocks;
               python - <<PY
import json
for m in ("rows", "blocks"):
    try:
        d = json.load(open("$O/shards8_%s.out" % m)); e = d["emulated_shard"]
        print(m, "slowest", round(e["slowest_ms"], 4), [(round(t["leaf_gb"], 2), round(t["ms_per_apply"], 4)) for t in e["all_ranks"]])
    except Exception as ex:
        print(m, "failed", ex)
PY
             }
shards8sum() { step shards8_rowsum 600 $B --emulate-world 8 --emulate-rank -1 --steps 20 --no-extra --no-cpu-baseline --shard rowsum;
               python - <<PY
import json
d = json.load(open("$O/shards8_rowsum.out")); e = d["emulated_shard"]
print("rowsum slowest", round(e["slowest_ms"], 4), [(round(t["leaf_gb"], 2), round(t["ms_per_apply"], 4)) for t in e["all_ranks"]])
PY
             }
shard0()     { step shard0_rows 300 $B --emulate-world 8 --emulate-rank 3 --steps 50 --no-extra --no-cpu-baseline --shard rows; grep "stage " $O/shard0_rows.err; }
lanes()      { step lanes_n65536 300 python tools/exp_lanes.py --n 65536 --lanes 2 3 4; cat $O/lanes_n65536.out;
               step lanes_shard8 400 python tools/exp_lanes.py --n 262144 --world 8 --lanes 2 3; cat $O/lanes_shard8.out; }
headline()   { step bench_default 900 $B --steps 20 --warmup 5 "$@"; cat $O/bench_default.out | cut -c1-1500; grep "stage " $O/bench_default.err; }
newtests()   { step pytest_new 900 python -m pytest tests -m gpu -x -q -k "shared_row_ranges or sphere or rccl or sharded"; tail -4 $O/pytest_new.out; }
flow()       { step flow_debug 120 python tools/debug_flow.py; tail -8 $O/flow_debug.out; tail -4 $O/flow_debug.err;
               step flow_test 300 python -m pytest tests -m gpu -x -q -k "dependency_driven or stage_profile"; tail -3 $O/flow_test.out;
               BFHIP_FLOW=1 step flow_n65536 200 $B --npoints 65536 --steps 50 --warmup 5 --no-cpu-baseline --no-extra; grep "stage " $O/flow_n65536.err;
               BFHIP_FLOW=1 step flow_shard 200 $B --emulate-world 8 --emulate-rank 3 --shard blocks --steps 50 --no-extra --no-cpu-baseline; grep "stage " $O/flow_shard.err;
               BFHIP_FLOW=1 step flow_head 200 $B --steps 20 --no-extra --no-cpu-baseline; grep "stage " $O/flow_head.err; }
streamshards() { step st_shards8 600 $B --workload streamer --emulate-world 8 --emulate-rank -1 --steps 20 --no-extra --no-cpu-baseline;
               python - <<PY
import json
d = json.load(open("$O/st_shards8.out")); e = d["emulated_shard"]
print("streamer rows slowest", round(e["slowest_ms"], 4), "full", round(d["ms_per_step"], 4), [(round(t["leaf_gb"], 2), round(t["ms_per_apply"], 4)) for t in e["all_ranks"]])
PY
               step st_test 600 python -m pytest tests -m gpu -x -q -k "shards_by_row_ranges"; tail -3 $O/st_test.out; }
stadj()      { step pytest_T 900 python -m pytest tests -m gpu -x -q -k "transpose or adjoint or streamer or rmul"; tail -3 $O/pytest_T.out;
               step st_adj 600 $B --workload streamer --adjoint --steps 10 --warmup 2 --no-cpu-baseline --no-extra;
               python - <<PY
import json
d = json.load(open("$O/st_adj.out")); print("streamer", d["value"], d["ms_per_step"], d["roofline"]["frac"], "adjoint", d["adjoint"])
PY
               grep "stage " $O/st_adj.err | tail -24; }
traceT()     { ( cd /tmp && export TMPDIR=/tmp && timeout -k 10 600 rocprofv3 --kernel-trace --output-format csv -d $O/traceT -- python3 $R/bench.py --workload streamer --adjoint --steps 3 --warmup 1 --no-cpu-baseline --no-extra > $O/traceT.json 2> $O/traceT.log ); echo "traceT exit=$?";
               python tools/trace_apply.py $O/traceT 120 > $O/traceT.txt; find $O/traceT -name "*.csv" -size +2M -delete; tail -60 $O/traceT.txt; }
streamer()   { step st_fwd 600 $B --workload streamer --adjoint --steps 10 --warmup 2 --no-cpu-baseline;
               python - <<PY
import json
d = json.load(open("$O/st_fwd.out")); print("streamer", d["value"], d["ms_per_step"], d["roofline"]["frac"], "adjoint", d["adjoint"]["ms_per_apply"], d["adjoint"]["frac_of_hbm_peak"], "cov", d.get("cov_matvec"))
PY
               grep "stage " $O/st_fwd.err | tail -12; }
three()      { step t_shard3 300 $B --emulate-world 8 --emulate-rank 3 --steps 50 --no-extra --no-cpu-baseline --shard rows
               python -c "import json; d = json.load(open('$O/t_shard3.out')); print('shard 3/8', d['ms_per_step'], d['roofline']['frac'])"
               step t_n65536 200 $B --npoints 65536 --steps 50 --warmup 5 --no-cpu-baseline --no-extra
               python -c "import json; d = json.load(open('$O/t_n65536.out')); print('N=65536', d['ms_per_step'], d['roofline']['frac'])"
               step t_head 300 $B --steps 20 --warmup 3 --no-cpu-baseline --no-extra
               python -c "import json; d = json.load(open('$O/t_head.out')); print('headline', d['ms_per_step'], d['roofline']['frac'])"; }
c128tests()  { step pytest_c 900 python -m pytest tests -m gpu -x -q -k "parity or fullsize or config4 or gmres or decor"; tail -3 $O/pytest_c.out; }
probe()      { step mfma_probe 300 tools/mfma_probe.bin; cat $O/mfma_probe.out; }
diag64()     { for v in base NOA NOX NOAX; do
                 if [ $v = base ]; then unset BFHIP_LIB_PATH; else export BFHIP_LIB_PATH=$R/butterfly_amd/csrc/exp/libbfhip_$v.so; fi
                 step r64_$v 300 $B --nrhs 64 --steps 5 --warmup 2 --no-cpu-baseline --no-extra
                 python -c "import json; d = json.load(open('$O/r64_$v.out')); print('$v', d['ms_per_step'], d['roofline']['achieved'])"; grep "stage " $O/r64_$v.err
               done; unset BFHIP_LIB_PATH; }
r64()        { step pytest_r64 600 python -m pytest tests -m gpu -x -q -k "rhs_block or dropin or nested or fullsize or linear or random"; tail -5 $O/pytest_r64.out;
               step r64_n65536 300 $B --npoints 65536 --nrhs 64 --steps 10 --warmup 2 --no-cpu-baseline --no-extra
               python -c "import json; d = json.load(open('$O/r64_n65536.out')); print('n65536', d['ms_per_step'], d['roofline']['achieved'], d['roofline']['frac'])"; grep "stage " $O/r64_n65536.err
               step r64_head 300 $B --nrhs 64 --steps 5 --warmup 2 --no-extra
               python -c "import json; d = json.load(open('$O/r64_head.out')); print('n262144', d['ms_per_step'], d['roofline']['achieved'], d['roofline']['frac'], d['cpu_baseline'].get('parity_rel_l2'))"; grep "stage " $O/r64_head.err; }
xcd()        { for v in ${XCD_VARIANTS:-base X4 X8 X16}; do
                 if [ $v = base ]; then unset BFHIP_LIB_PATH; else export BFHIP_LIB_PATH=$R/butterfly_amd/csrc/exp/libbfhip_$v.so; fi
                 step x64_$v 300 $B --nrhs 64 --steps 5 --warmup 2 --no-cpu-baseline --no-extra
                 python -c "import json; d = json.load(open('$O/x64_$v.out')); print('$v', d['ms_per_step'], d['roofline']['achieved'], d['roofline']['frac'])"; grep "stage " $O/x64_$v.err | tail -5
               done; unset BFHIP_LIB_PATH; }
pmc64()      { ( cd /tmp && export TMPDIR=/tmp
                 for v in ${PMC_VARIANTS:-base}; do
                   if [ $v = base ]; then unset BFHIP_LIB_PATH; else export BFHIP_LIB_PATH=$R/butterfly_amd/csrc/exp/libbfhip_$v.so; fi
                   timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch_$v -- python3 $R/bench.py --nrhs 64 --steps 2 --warmup 1 --no-cpu-baseline --no-extra > $O/fetch_$v.json 2> $O/fetch_$v.log; echo "fetch_$v exit=$?"
                   timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/mfma_$v -- python3 $R/bench.py --nrhs 64 --steps 2 --warmup 1 --no-cpu-baseline --no-extra > $O/mfma_$v.json 2> $O/mfma_$v.log; echo "mfma_$v exit=$?"
                   python3 $R/tools/pmc_quick.py $O/fetch_$v $O/mfma_$v
                 done ) }
r64small()   { for v in ${XCD_VARIANTS:-base}; do
                 if [ $v = base ]; then unset BFHIP_LIB_PATH; else export BFHIP_LIB_PATH=$R/butterfly_amd/csrc/exp/libbfhip_$v.so; fi
                 step s64_$v 300 $B --npoints 65536 --nrhs 64 --steps 10 --warmup 2 --no-cpu-baseline --no-extra
                 python -c "import json; d = json.load(open('$O/s64_$v.out')); print('n65536 $v', d['ms_per_step'], d['roofline']['achieved'], d['roofline']['frac'])"
               done; unset BFHIP_LIB_PATH; }
inline64()   { step def1 600 $B --steps 20 --warmup 5 --no-streamer
               python -c "import json; d = json.load(open('$O/def1.out')); print('default line: headline', d['roofline']['frac'], 'nrhs64', d['nrhs64']['roofline']['frac'], d['nrhs64']['ms_per_apply'], 'n65536', d['n65536']['roofline']['frac'])"
               step alone64 300 $B --nrhs 64 --steps 10 --warmup 3 --no-cpu-baseline --no-extra
               python -c "import json; d = json.load(open('$O/alone64.out')); print('standalone nrhs64', d['ms_per_step'], d['roofline']['frac'])"
               step alone64b 300 $B --nrhs 64 --steps 30 --warmup 10 --no-cpu-baseline --no-extra
               python -c "import json; d = json.load(open('$O/alone64b.out')); print('standalone nrhs64 (30 steps)', d['ms_per_step'], d['roofline']['frac'])"; }
adjab()      { for v in ${ADJ_VARIANTS:-T0 base}; do
                 if [ $v = base ]; then unset BFHIP_LIB_PATH; else export BFHIP_LIB_PATH=$R/butterfly_amd/csrc/exp/libbfhip_$v.so; fi
                 step adj_$v 300 $B --adjoint --steps 10 --warmup 2 --no-cpu-baseline --no-extra
                 python -c "import json; d = json.load(open('$O/adj_$v.out')); print('$v c128 adjoint', d['adjoint']['ms_per_apply'], d['adjoint']['frac_of_hbm_peak'], d['adjoint']['transpose_identity_rel'])"
                 step adjs_$v 300 $B --npoints 65536 --adjoint --steps 30 --warmup 3 --no-cpu-baseline --no-extra
                 python -c "import json; d = json.load(open('$O/adjs_$v.out')); print('$v n65536 adjoint', d['adjoint']['ms_per_apply'], d['adjoint']['frac_of_hbm_peak'])"
                 step stadj_$v 400 $B --workload streamer --adjoint --steps 10 --warmup 2 --no-cpu-baseline --no-extra
                 python -c "import json; d = json.load(open('$O/stadj_$v.out')); print('$v streamer fwd', d['ms_per_step'], d['roofline']['frac'], 'adjoint', d['adjoint']['ms_per_apply'], d['adjoint']['frac_of_hbm_peak'], d['adjoint']['transpose_identity_rel'])"
               done; unset BFHIP_LIB_PATH; }
packed()     { step pytest_p 600 python -m pytest tests -m gpu -x -q -k "adjoint or transpose or rmul or cov or save_load"; tail -3 $O/pytest_p.out
               for v in "" "--adjoint-shared"; do t=$(echo "x$v" | tr -d ' -');
                 step adj_$t 300 $B --adjoint $v --steps 10 --warmup 2 --no-cpu-baseline --no-extra
                 python -c "import json; d = json.load(open('$O/adj_$t.out')); print('$t c128 adjoint', d['adjoint']['ms_per_apply'], d['adjoint']['frac_of_hbm_peak'], d['adjoint']['transpose_identity_rel'], 'fwd', d['ms_per_step'])"
                 step adjs_$t 300 $B --npoints 65536 --adjoint $v --steps 30 --warmup 3 --no-cpu-baseline --no-extra
                 python -c "import json; d = json.load(open('$O/adjs_$t.out')); print('$t n65536 adjoint', d['adjoint']['ms_per_apply'], d['adjoint']['frac_of_hbm_peak'])"
                 step stadj_$t 400 $B --workload streamer --adjoint $v --steps 10 --warmup 2 --no-cpu-baseline --no-extra
                 python -c "import json; d = json.load(open('$O/stadj_$t.out')); print('$t streamer fwd', d['ms_per_step'], d['roofline']['frac'], 'adjoint', d['adjoint']['ms_per_apply'], d['adjoint']['frac_of_hbm_peak'], d['adjoint']['transpose_identity_rel'], 'cov', d['cov_matvec']['ms_per_product'], d['cov_matvec']['rel_vs_separate_applies'])"
               done; }
stpacked()   { step pytest_st 900 python -m pytest tests -m gpu -x -q -k "real or nested or streamer or adjoint or transpose or rmul or cov or packed"; tail -3 $O/pytest_st.out
               for v in shared packed; do
                 if [ $v = shared ]; then X=--adjoint-shared; else X=; fi
                 step stadj_$v 400 $B --workload streamer --adjoint $X --steps 10 --warmup 2 --no-cpu-baseline --no-extra
                 python -c "import json; d = json.load(open('$O/stadj_$v.out')); print('$v streamer fwd', d['ms_per_step'], d['roofline']['frac'], 'adjoint', d['adjoint']['ms_per_apply'], d['adjoint']['frac_of_hbm_peak'], d['adjoint']['transpose_identity_rel'], 'cov', d['cov_matvec']['ms_per_product'], d['cov_matvec']['rel_vs_separate_applies'])"
               done; }
libab()      { for v in ${LIB_VARIANTS:-base}; do
                 if [ $v = base ]; then unset BFHIP_LIB_PATH; else export BFHIP_LIB_PATH=$R/butterfly_amd/csrc/exp/libbfhip_$v.so; fi
                 step st_$v 400 $B --workload streamer --adjoint --steps 10 --warmup 2 --no-cpu-baseline --no-extra
                 python -c "import json; d = json.load(open('$O/st_$v.out')); print('$v streamer fwd', d['ms_per_step'], d['roofline']['frac'], 'adjoint', d['adjoint']['ms_per_apply'], d['adjoint']['frac_of_hbm_peak'], d['adjoint']['transpose_identity_rel'])"
                 grep "stage " $O/st_$v.err | head -9
               done; unset BFHIP_LIB_PATH; }
tracePk()    { ( cd /tmp && export TMPDIR=/tmp && timeout -k 10 600 rocprofv3 --kernel-trace --output-format csv -d $O/tracePk -- python3 $R/bench.py --workload streamer --adjoint --steps 2 --warmup 1 --no-cpu-baseline --no-extra > $O/tracePk.json 2> $O/tracePk.log ); echo "tracePk exit=$?";
               python tools/trace_apply.py $O/tracePk 70 > $O/tracePk.txt; find $O/tracePk -name "*.csv" -size +2M -delete; grep -v "at::native\|rocclr" $O/tracePk.txt | tail -45; }
rehearse()   { step rh_rccl 300 $B --force-collective --steps 10 --warmup 2 --no-cpu-baseline --no-extra
               python -c "import json; d = json.load(open('$O/rh_rccl.out')); m = d['multi_gpu']; print('rccl', d['ms_per_step'], m['collective_impl'], m['ranks_agree'], m['max_local_ms'], m['max_collective_ms'])"
               BENCH_FORCE_TORCH_COLLECTIVE=1 step rh_torch 300 $B --force-collective --steps 10 --warmup 2 --no-cpu-baseline --no-extra
               python -c "import json; d = json.load(open('$O/rh_torch.out')); m = d['multi_gpu']; print('torch', d['ms_per_step'], m['collective_impl'], m['collective_fallback_reason'], m['ranks_agree'], m['max_local_ms'], m['max_collective_ms'])"
               BENCH_ALSO_TIME_ROWS=1 step rh_rowsum 400 $B --force-collective --shard rowsum --steps 10 --warmup 2 --cpu-budget-gb 7
               python -c "import json; d = json.load(open('$O/rh_rowsum.out')); m = d['multi_gpu']; print('rowsum', d['ms_per_step'], m['mode'], m['ranks_agree'], 'also', m.get('also_timed'), 'parity', d['cpu_baseline'].get('parity_rel_l2'), d['cpu_baseline'].get('parity_covers_shared_block_row'))"
               BENCH_ALSO_TIME_ROWS=1 BENCH_FORCE_TORCH_COLLECTIVE=1 step rh_blocks 400 $B --force-collective --shard blocks --steps 10 --warmup 2 --no-cpu-baseline --no-extra
               python -c "import json; d = json.load(open('$O/rh_blocks.out')); m = d['multi_gpu']; print('blocks/torch', d['ms_per_step'], m['mode'], m['collective_impl'], m['ranks_agree'], 'also', m.get('also_timed'))"; }
flowab()     { for v in ${FLOW_VARIANTS:-exp_static exp_tickets}; do export BFHIP_LIB_PATH=$R/butterfly_amd/csrc/exp/libbfhip_$v.so
                 if [ $v = exp_static ]; then step flowtest_$v 600 python -m pytest tests/experimental_checks.py -m gpu -x -q -k "dependency_driven and not 65536" -p no:cacheprovider; tail -2 $O/flowtest_$v.out; fi
                 BFHIP_FLOW=1 step fl_n65536_$v 200 $B --npoints 65536 --steps 50 --warmup 5 --no-cpu-baseline --no-extra
                 python -c "import json; d = json.load(open('$O/fl_n65536_$v.out')); print('$v n65536 flow', d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernel'])"
                 BFHIP_FLOW=1 step fl_shard_$v 200 $B --emulate-world 8 --emulate-rank 3 --shard rows --steps 50 --no-extra --no-cpu-baseline
                 python -c "import json; d = json.load(open('$O/fl_shard_$v.out')); print('$v shard3 flow', d['ms_per_step'], d['roofline']['frac'])"
                 BFHIP_FLOW=1 step fl_head_$v 200 $B --steps 20 --no-extra --no-cpu-baseline
                 python -c "import json; d = json.load(open('$O/fl_head_$v.out')); print('$v headline flow', d['ms_per_step'], d['roofline']['frac'])"
               done; unset BFHIP_LIB_PATH
               step st_n65536 200 $B --npoints 65536 --steps 50 --warmup 5 --no-cpu-baseline --no-extra
               python -c "import json; d = json.load(open('$O/st_n65536.out')); print('staged n65536', d['ms_per_step'], d['roofline']['frac'])"
               step st_shard 200 $B --emulate-world 8 --emulate-rank 3 --shard rows --steps 50 --no-extra --no-cpu-baseline
               python -c "import json; d = json.load(open('$O/st_shard.out')); print('staged shard3', d['ms_per_step'], d['roofline']['frac'])"; }
rhssweep()   { for q in 1 2 3 4 8 16 32 48 64; do
                 step rhs_$q 200 $B --nrhs $q --steps 5 --warmup 2 --no-cpu-baseline --no-extra
                 python -c "import json; d = json.load(open('$O/rhs_$q.out')); print('nrhs $q', round(d['ms_per_step'], 3), 'ms', d['roofline']['kernel'], round(d['roofline']['frac'], 3))"
               done; }
loopprobe()  { step loop_probe 200 tools/mfma_loop_probe.bin; cat $O/loop_probe.out
               ( cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_loop -- $R/tools/mfma_loop_probe.bin 300 > $O/pmc_loop.out 2> $O/pmc_loop.log; echo "pmc_loop exit=$?"
                 PMC_QUICK_PER_DISPATCH=1 PMC_QUICK_FILTER=probeLoop python3 $R/tools/pmc_quick.py $O/pmc_loop ) }
dma1()       { step loop_probe 300 tools/mfma_loop_probe.bin 600; cat $O/loop_probe.out
               step pytest_dma 900 python -m pytest tests -m gpu -x -q -k "rhs_block or sharded or rccl or plain_c or gmres or dropin or fullsize"; tail -5 $O/pytest_dma.out
               step r64_head 300 $B --nrhs 64 --steps 10 --warmup 3 --no-extra --no-cpu-baseline
               python -c "import json; d = json.load(open('$O/r64_head.out')); print('n262144 nrhs64', d['ms_per_step'], d['roofline']['achieved'], d['roofline']['frac'])"; grep "stage " $O/r64_head.err; }
pmcloop()    { step loop_probe 300 tools/mfma_loop_probe.bin 600; cat $O/loop_probe.out
               ( cd /tmp && export TMPDIR=/tmp && timeout -k 10 400 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_loop -- $R/tools/mfma_loop_probe.bin 400 > $O/pmc_loop.out 2> $O/pmc_loop.log; echo "pmc_loop exit=$?"
                 PMC_QUICK_PER_DISPATCH=1 PMC_QUICK_FILTER=probeLoop python3 $R/tools/pmc_quick.py $O/pmc_loop
                 timeout -k 10 400 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_loop2 -- $R/tools/mfma_loop_probe.bin 400 > $O/pmc_loop2.out 2> $O/pmc_loop2.log; echo "pmc_loop2 exit=$?"
                 find $O/pmc_loop $O/pmc_loop2 -name "*.csv" -size +4M -delete ) }
sharedx()    { step sharedx_probe 300 tools/mfma_sharedx_probe.bin 600; cat $O/sharedx_probe.out
               ( cd /tmp && export TMPDIR=/tmp && timeout -k 10 400 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_sx -- $R/tools/mfma_sharedx_probe.bin 400 > $O/pmc_sx.out 2> $O/pmc_sx.log; echo "pmc_sx exit=$?"
                 PMC_QUICK_PER_DISPATCH=1 PMC_QUICK_FILTER=probe python3 $R/tools/pmc_quick.py $O/pmc_sx
                 timeout -k 10 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_sx_fetch -- $R/tools/mfma_sharedx_probe.bin 400 > $O/pmc_sx_fetch.out 2> $O/pmc_sx_fetch.log; echo "pmc_sx_fetch exit=$?"
                 python3 - <<PY
import csv, glob, collections
cc = sorted(glob.glob("$O/pmc_sx_fetch/*/*_counter_collection.csv"))[-1]
per = collections.defaultdict(float); nm = {}
for r in csv.DictReader(open(cc)):
    if r["Counter_Name"] == "FETCH_SIZE": per[r["Dispatch_Id"]] += float(r["Counter_Value"]); nm[r["Dispatch_Id"]] = r["Kernel_Name"][:30]
for d in sorted(per, key=int): print("dispatch", d, nm[d], "fetch GB %.2f" % (per[d] * 1024 * 2 / 1e9))
PY
                 find $O/pmc_sx $O/pmc_sx_fetch -name "*.csv" -size +4M -delete ) }
r5tests()    { step pytest_r5 1100 python -m pytest tests/test_gpu_config4.py tests/test_gpu_config4_rowsum.py tests/test_bench_gpu.py tests/test_gpu_sharded_adjoint.py -m gpu -x -q; tail -6 $O/pytest_r5.out; }
pcie()       { step pcie_head 400 $B --steps 20 --warmup 3 --pcie --no-cpu-baseline --no-extra
               python -c "import json; d = json.load(open('$O/pcie_head.out')); print('headline', d['ms_per_step'], d['pcie_inclusive'])"
               step pcie_n65536 300 $B --npoints 65536 --steps 50 --warmup 5 --pcie --no-cpu-baseline --no-extra
               python -c "import json; d = json.load(open('$O/pcie_n65536.out')); print('n65536', d['ms_per_step'], d['pcie_inclusive'])"; }
defline()    { local t0=$(date +%s); step def_line 900 $B; echo "default line wall seconds: $(( $(date +%s) - t0 ))"
               python - <<PY
import json
d = json.load(open("$O/def_line.out"))
print("headline", d["value"], d["ms_per_step"], d["roofline"]["frac"], "resident", d["config"].get("resident_bytes"))
print("cpu_baseline", d["cpu_baseline"]["value"], d["cpu_baseline"]["sample"][:160], d["cpu_baseline"].get("parity_rel_l2"))
print("adjoint", d["adjoint"]["frac_of_hbm_peak"], "shared", d.get("adjoint_shared"))
print("pcie", {k: v for k, v in d["pcie_inclusive"].items() if k != "note"})
n = d["nrhs64"]; print("nrhs64", n.get("ms_per_apply"), n.get("roofline", {}).get("frac"), n.get("whole_apply_frac"), n.get("cpu_baseline", {}).get("value"), n.get("cpu_baseline", {}).get("all_cores"), n.get("error"))
m = d["n65536"]; print("n65536", m.get("ms_per_step"), m.get("roofline", {}).get("frac"), (m.get("cpu_baseline") or {}).get("sample", "")[:100], {k: v for k, v in (m.get("pcie_inclusive") or {}).items() if k != "note"}, m.get("error"))
c = d["configs4_streamer"]; print("streamer", c.get("ms_per_step"), c.get("roofline", {}).get("frac"), "adj", (c.get("adjoint") or {}).get("frac_of_hbm_peak"), "shared", (c.get("adjoint_shared") or {}).get("frac_of_hbm_peak"), (c.get("cpu_baseline") or {}).get("sample", "")[:120], c.get("error"))
PY
             }
for s in "$@"; do $s; done
