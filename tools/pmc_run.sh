#!/bin/bash
# usage: tools/pmc_run.sh <tag> <kernel-substring> -- <command...>
# One rocprofv3 --pmc pass per counter group (no tracing flags), summaries into
# gpurun_out/pmc_<tag>.txt
tag=$1; sub=$2; shift 3
repo=$(cd "$(dirname "$0")/.." && pwd)
out=$repo/gpurun_out/pmc_$tag
mkdir -p "$out"
export TMPDIR=/tmp
groups=(
 "SQ_WAVES GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES"
 "SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY"
 "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU"
 "SQ_INSTS_MFMA SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAIT_INST_LDS"
 "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM"
 "TA_TA_BUSY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum"
 "TCC_HIT_sum TCC_MISS_sum"
)
i=0
files=()
for g in "${groups[@]}"; do
  (cd /tmp && rocprofv3 --pmc $g --output-format csv -d "$out/g$i" -o pass -- "$@" > "$out/g$i.log" 2>&1)
  f=$(find "$out/g$i" -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && files+=("$f")
  i=$((i+1))
done
python "$repo/tools/pmc_summary.py" "$sub" "${files[@]}" | tee "$repo/gpurun_out/pmc_$tag.txt"
rm -rf "$out"
