#!/bin/bash
# Collects the rocprofv3 evidence committed under profiles/ (run on the GPU box via gpurun).
#   usage: tools/profile_round.sh <round-tag> [helm2|streamer|streamerT|all]      e.g. r2 helm2
# (every pass lays its operand out and synthesizes it again: ~25 s for the headline operand, ~40 s for the streamed one)
# Each PMC set is its own pass with --kernel-trace only (no --stats / sys-trace with --pmc).
set -u
TAG=${1:-r1}
PART=${2:-all}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() { # name, rocprof args..., -- bench args
  local name=$1; shift
  local rp=(); while [ "$1" != "--" ]; do rp+=("$1"); shift; done; shift
  rocprofv3 "${rp[@]}" --output-format csv -d $OUT/$name -- python3 $R/bench.py "$@" --no-cpu-baseline --no-extra > $OUT/$name.json 2> $OUT/$name.log
  echo "$name exit=$?"
}
if [ "$PART" != "streamer" ] && [ "$PART" != "streamerT" ]; then
# headline: N=262144, nrhs=1
run stats_r1   --kernel-trace --stats -- --steps 10 --warmup 2
run fetch_r1   --kernel-trace --pmc FETCH_SIZE -- --steps 3 --warmup 1
run write_r1   --kernel-trace --pmc WRITE_SIZE -- --steps 3 --warmup 1
run lds_r1     --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES GRBM_GUI_ACTIVE -- --steps 3 --warmup 1
# 64 right-hand sides (MFMA kernel)
run stats_r64  --kernel-trace --stats -- --nrhs 64 --steps 3 --warmup 1
run mfma_r64   --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -- --nrhs 64 --steps 2 --warmup 1
run fetch_r64  --kernel-trace --pmc FETCH_SIZE -- --nrhs 64 --steps 2 --warmup 1
run write_r64  --kernel-trace --pmc WRITE_SIZE -- --nrhs 64 --steps 2 --warmup 1
# BASELINE configs[1]: N = 65536, one right-hand side (short launches)
run stats_n65536 --kernel-trace --stats -- --npoints 65536 --steps 20 --warmup 3
run fetch_n65536 --kernel-trace --pmc FETCH_SIZE -- --npoints 65536 --steps 5 --warmup 1
run write_n65536 --kernel-trace --pmc WRITE_SIZE -- --npoints 65536 --steps 5 --warmup 1
# adjoint apply of the headline operand on the shared leaves (bfStageKernelT; bench.py's default, a packed copy for A^T, runs the forward kernels)
run stats_adj  --kernel-trace --stats -- --adjoint --adjoint-shared --steps 5 --warmup 1
run fetch_adj  --kernel-trace --pmc FETCH_SIZE -- --adjoint --adjoint-shared --steps 3 --warmup 1
run write_adj  --kernel-trace --pmc WRITE_SIZE -- --adjoint --adjoint-shared --steps 3 --warmup 1
fi
SQSET="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"
if [ "$PART" != "helm2" ] && [ "$PART" != "streamerT" ]; then
# BASELINE configs[4]: streamed real butterfly, N = 1M fp32 (bfStageKernelReal<f32>)
run stats_st   --kernel-trace --stats -- --workload streamer --steps 5 --warmup 1
run fetch_st   --kernel-trace --pmc FETCH_SIZE -- --workload streamer --steps 2 --warmup 1
run write_st   --kernel-trace --pmc WRITE_SIZE -- --workload streamer --steps 2 --warmup 1
run sq_st      --kernel-trace --pmc $SQSET -- --workload streamer --steps 2 --warmup 1
fi
if [ "$PART" != "helm2" ]; then
# its transposed apply on the shared leaves (bfStageKernelT<f32>: the 16- and the 64-column tiling of a stage in one launch; bench.py's
# default, a packed copy for A^T, runs the forward kernels)
run stats_stT  --kernel-trace --stats -- --workload streamer --adjoint --adjoint-shared --steps 5 --warmup 1
run fetch_stT  --kernel-trace --pmc FETCH_SIZE -- --workload streamer --adjoint --adjoint-shared --steps 2 --warmup 1
run write_stT  --kernel-trace --pmc WRITE_SIZE -- --workload streamer --adjoint --adjoint-shared --steps 2 --warmup 1
# where the wavefronts' cycles go (parked on memory / issuing): the transposed kernels next to the forward ones in the same pass
run sq_stT     --kernel-trace --pmc $SQSET -- --workload streamer --adjoint --adjoint-shared --steps 2 --warmup 1
fi
ls $OUT
