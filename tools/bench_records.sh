#!/bin/bash
# Regenerates the bench / builder / ceiling records committed under profiles/ (run on the GPU box):
#   gpurun --timeout 1200 -- 'bash tools/bench_records.sh r1'
# PMC and kernel-stats summaries come from tools/profile_round.sh + tools/summarize_profiles.py.
set -u
TAG=${1:-r1}
PART=${2:-all}        # all | single | shards | streamer
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/records_$TAG
mkdir -p $O
cd $R
B="timeout -k 10 400 python bench.py"
run() { local name=$1; shift; "$@" > $O/$name.json 2> $O/$name.err; echo "$name exit=$?"; }      # (an env assignment in front of `run` reaches the command)
if [ "$PART" = all ] || [ "$PART" = single ]; then
run ${TAG}_bench_n262144                 $B --pcie
run ${TAG}_bench_n262144_nrhs64          $B --nrhs 64
run ${TAG}_bench_n65536                  $B --npoints 65536
run ${TAG}_bench_n65536_nrhs64           $B --npoints 65536 --nrhs 64 --no-cpu-baseline
run ${TAG}_bench_n262144_adjoint         $B --adjoint --no-cpu-baseline --no-extra
run ${TAG}_bench_n262144_adjoint_shared  $B --adjoint --adjoint-shared --no-cpu-baseline --no-extra
run ${TAG}_bench_n262144_rccl_1rank      $B --force-collective --no-cpu-baseline --no-extra
fi
if [ "$PART" = all ] || [ "$PART" = shards ]; then
# every rank's shard of a 2 / 4 / 8-GPU job, one after another on this GPU (the slowest bounds the job)
for w in 2 4 8; do run ${TAG}_bench_n262144_shards${w}_emulated $B --emulate-world $w --emulate-rank -1 --steps 20 --no-cpu-baseline --no-extra; done
for m in rows rowsum blocks; do run ${TAG}_bench_n262144_shards8_${m}_emulated $B --emulate-world 8 --emulate-rank -1 --steps 20 --shard $m --no-cpu-baseline --no-extra; done
# ... and every rank's part of the sharded ADJOINT step (A_r^T on its rows of v, shared leaves), before the one all-reduce
for m in rows rowsum; do run ${TAG}_bench_n262144_shards8_${m}_adjoint_emulated $B --emulate-world 8 --emulate-rank -1 --steps 20 --shard $m --adjoint --no-cpu-baseline --no-extra; done
run ${TAG}_bench_n1048576_shards8_emulated    $B --npoints 1048576 --emulate-world 8 --emulate-rank -1 --steps 10 --no-cpu-baseline --no-extra
fi
if [ "$PART" = all ] || [ "$PART" = streamer ]; then
# the experimental one-launch executor (BFHIP_FLAG_FLOW) and the persistent ticket launch next to the staged launches
# (they live in libbfhip_exp.so only: make -C butterfly_amd/csrc experimental)
EXPLIB=$(pwd)/butterfly_amd/csrc/libbfhip_exp.so
BFHIP_LIB_PATH=$EXPLIB BFHIP_FLOW=1 run ${TAG}_bench_n65536_flow $B --npoints 65536 --no-cpu-baseline --no-extra
BFHIP_LIB_PATH=$EXPLIB BFHIP_FLOW=1 run ${TAG}_bench_n262144_flow $B --no-cpu-baseline --no-extra
BFHIP_LIB_PATH=$EXPLIB BFHIP_PERSISTENT=1 run ${TAG}_bench_n65536_persistent $B --npoints 65536 --no-cpu-baseline --no-extra
BFHIP_LIB_PATH=$EXPLIB BFHIP_PERSISTENT=1 run ${TAG}_bench_n262144_persistent $B --no-cpu-baseline --no-extra
# BASELINE configs[4]: the streamed real butterfly (fac_streamer structure, rank model), fp32 and fp64
S="timeout -k 10 900 python bench.py --workload streamer"
run ${TAG}_bench_streamer_n1048576_f32   $S --adjoint --steps 10
run ${TAG}_bench_streamer_n1048576_f32_adjoint_shared $S --adjoint --adjoint-shared --steps 10 --no-cpu-baseline
run ${TAG}_bench_streamer_n1048576_f64   $S --dtype f64 --adjoint --steps 10 --no-cpu-baseline
run ${TAG}_bench_streamer_n262144_f32    $S --npoints 262144 --lmax 127 --adjoint --steps 10
run ${TAG}_build_n65536                  timeout -k 10 400 python tools/build_fullsize.py --npoints 65536
run ${TAG}_build_n262144                 timeout -k 10 400 python tools/build_fullsize.py --npoints 262144
run ${TAG}_bie_device_n65536_k100        timeout -k 10 400 python tools/helm2_bie_device.py --npoints 65536 --wavenumber 100 --max-iter 300
hipcc -O3 --offload-arch=gfx950 tools/hbm_peak.hip -o /tmp/hbm_peak 2>/dev/null && run ${TAG}_hbm_peak timeout -k 5 120 /tmp/hbm_peak
hipcc -O3 --offload-arch=gfx950 tools/mfma_peak.hip -o /tmp/mfma_peak 2>/dev/null && run ${TAG}_mfma_peak timeout -k 5 120 /tmp/mfma_peak
hipcc -O3 --offload-arch=gfx950 tools/launch_floor.hip -o /tmp/launch_floor 2>/dev/null && run ${TAG}_launch_floor timeout -k 5 200 /tmp/launch_floor
run ${TAG}_timeline_n65536               timeout -k 10 300 python tools/timeline.py --npoints 65536
run ${TAG}_timeline_shard3of8            timeout -k 10 300 python tools/timeline.py --npoints 262144 --world 8 --rank 3
fi
ls $O
