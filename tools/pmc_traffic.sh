#!/bin/bash
# usage: tools/pmc_traffic.sh <tag> <kernel-substring> -- <command...>
# HBM-side traffic of one kernel: FETCH_SIZE and WRITE_SIZE in SEPARATE
# rocprofv3 --pmc passes (they do not fit one pass, MI355X_MICROARCH.md), no
# tracing flags.  Summary -> gpurun_out/pmc_traffic_<tag>.txt
tag=$1; sub=$2; shift 3
repo=$(cd "$(dirname "$0")/.." && pwd)
out=$repo/gpurun_out/pmct_$tag
mkdir -p "$out"
export TMPDIR=/tmp
files=()
i=0
for g in "FETCH_SIZE" "WRITE_SIZE"; do
  (cd /tmp && rocprofv3 --pmc $g --output-format csv -d "$out/g$i" -o pass -- "$@" > "$out/g$i.log" 2>&1)
  f=$(find "$out/g$i" -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && files+=("$f")
  i=$((i+1))
done
python "$repo/tools/pmc_summary.py" "$sub" "${files[@]}" | tee "$repo/gpurun_out/pmc_traffic_$tag.txt"
rm -rf "$out"
