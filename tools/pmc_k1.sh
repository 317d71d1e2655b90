#!/bin/bash
# Hardware counters of the projection kernel on configuration 3 (GPU box).  One rocprofv3 pass per
# counter group (--pmc is never combined with the trace domains gpurun refuses); per-kernel sums
# are written to gpurun_out/pmc_<tag>.json by tools/pmc_summarise.py.
#   tools/pmc_k1.sh <tag> [bench.py args...]
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
tag=$1; shift
out=$ROOT/gpurun_out/pmc_$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
groups=(
 "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT"
 "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS"
 "SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_RD SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INST_LEVEL_VMEM SQ_VALU_MFMA_COEXEC_CYCLES SQ_INST_CYCLES_SALU"
 "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum"
 "TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_TCP_LATENCY_sum SQ_IFETCH"
)
i=0
for g in "${groups[@]}"; do
  rocprofv3 --kernel-trace --pmc $g --output-format csv -d "$out/g$i" -- python3 "$ROOT/bench.py" --steps 2 --warmup 1 --no-cpu-baseline "$@" > "$out/g$i.log" 2>&1 || echo "group $i failed (see $out/g$i.log)"
  i=$((i+1))
done
python3 "$ROOT/tools/pmc_summarise.py" "$out" > "$ROOT/gpurun_out/pmc_$tag.json"
cat "$ROOT/gpurun_out/pmc_$tag.json"
