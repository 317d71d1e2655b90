#!/bin/bash
# Hardware counters of the projection kernel on configuration 3 (GPU box).  One rocprofv3 pass per
# counter group (--pmc is never combined with the trace domains gpurun refuses; the TA_* / TCP_*
# counters hung rocprofv3 on this pool and are left out); per-kernel means are written to
# gpurun_out/pmc_<tag>.json by tools/pmc_summarise.py.
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
 "SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_RD SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INST_LEVEL_VMEM SQ_VALU_MFMA_COEXEC_CYCLES SQ_LDS_IDX_ACTIVE"
 "FETCH_SIZE"
 "WRITE_SIZE"
)
i=0
for g in "${groups[@]}"; do
  rocprofv3 --kernel-trace --pmc $g --output-format csv -d "$out/g$i" -- python3 "$ROOT/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-extras "$@" > "$out/g$i.log" 2>&1 || echo "group $i failed (see $out/g$i.log)"
  i=$((i+1))
done
python3 "$ROOT/tools/pmc_summarise.py" "$out" > "$ROOT/gpurun_out/pmc_$tag.json"
cat "$ROOT/gpurun_out/pmc_$tag.json"
