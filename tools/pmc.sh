#!/bin/bash
# Collects rocprofv3 PMC counters for the render kernel in separate passes
# (one counter group per run; never combined with sys/hip/hsa tracing).
#   tools/pmc.sh <tag> [bench args...]      -> gpurun_out/pmc_<tag>/<group>_counter_collection.csv
# Run ON the GPU box (via gpurun) from the repo root.
set -u
TAG=${1:-run}; shift || true
OUT=gpurun_out/pmc_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
ARGS="--steps 128 --warmup 64 --no-cpu-baseline $*"
GROUPS_TO_RUN=${PMC_GROUPS:-"sq1 sq2"}
declare -A G
G[sq1]="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY"
G[sq2]="SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INSTS_SALU SQ_INST_LEVEL_VMEM SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_LDS"
G[ta]="TA_BUSY_avr TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum"
G[tcc]="TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum"
G[fetch]="FETCH_SIZE GRBM_GUI_ACTIVE"
G[write]="WRITE_SIZE"
G[rdsize]="TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum"
G[tcp1]="TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum"
G[tcp2]="TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_TCC_READ_REQ_LATENCY_sum"
G[tcp3]="TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TOTAL_ACCESSES_sum TCP_TOTAL_READ_sum"
G[tcp4]="TCP_LFIFO_STALL_CYCLES_sum TCP_RFIFO_STALL_CYCLES_sum TCP_TCP_LATENCY_sum TCP_TCR_RDRET_STALL_sum"
# TA_* and TD_* counters abort rocprofv3 on this pool (measured twice): never select them
G[ta1]="TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum"
G[td1]="TD_TD_BUSY_sum TD_TC_STALL_sum TD_LOAD_WAVEFRONT_sum"
G[tlb]="TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_TCC_READ_REQ_LATENCY_sum"
for g in $GROUPS_TO_RUN; do
  ( cd /tmp && timeout ${PMC_TIMEOUT:-240} rocprofv3 --kernel-trace --pmc ${G[$g]} -d "$GRAFT_REPO_ROOT/$OUT" -o $g --output-format csv -- \
      python "$GRAFT_REPO_ROOT/bench.py" $ARGS > "$GRAFT_REPO_ROOT/$OUT/$g.json" 2> "$GRAFT_REPO_ROOT/$OUT/$g.log" ) || echo "pass $g failed"
done
python tools/pmc_summary.py "$OUT" | tee "$OUT/summary.txt"
