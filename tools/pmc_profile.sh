#!/bin/bash
# PMC counter passes for the integrator kernel (run on the MI355X box via gpurun).
#   tools/pmc_profile.sh <out_dir_under_gpurun_out> [bench.py args...]
# For per-launch figures every launch of a kernel should have the same shape: run it with PT_ESCAPE_AFTER=0 (escape masks from the
# first frame) and PT_QUEUE_GIB=80 (first frame in one pass) in the environment, e.g.
#   PT_ESCAPE_AFTER=0 PT_QUEUE_GIB=80 bash tools/pmc_profile.sh pmc_r04_j --steps 1 --warmup 2 --no-legacy
#   python tools/pmc_to_profiles.py gpurun_out/pmc_r04_j r04_j 500000 1920 1080 128 5 1 --scene-flags=8 --frames=5   (4 warm-up frames + 1)
# Each pass is its own rocprofv3 run with --pmc only (never combined with trace domains other than
# kernel-trace); tools/pmc_summarize.py folds the CSVs into one JSON for profiles/.
set -u
OUT=gpurun_out/$1; shift
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
PASSES=(
 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY"
 "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INSTS_FLAT"
 "FETCH_SIZE GRBM_GUI_ACTIVE"
 "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"
 "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_ACCESSES_sum"
 "SQ_INSTS_BRANCH SQ_INSTS_VALU_TRANS_F32 SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32"
)
i=0
for p in "${PASSES[@]}"; do
  i=$((i+1))
  timeout -k 10 400 rocprofv3 --pmc $p --output-format csv -d "$OUT/pass$i" -- python3 bench.py --cpu-seconds 0 --no-counters "$@" > "$OUT/pass$i.log" 2>&1 || { echo "pass $i failed"; tail -5 "$OUT/pass$i.log"; exit 1; }
  echo "pass $i ok: $p"
done
