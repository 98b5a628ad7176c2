#!/bin/bash
# A few PMC groups of one bench.py command (diagnostics while tuning; tools/pmc_collect.sh is the full set for profiles/).
# Usage on the GPU box: bash tools/pmc_quick.sh <tag> <kernel-substring> "<group1>" "<group2>" ... -- <bench.py args>
set -e
TAG=$1; KERN=$2; shift 2
GR=()
while [ "$1" != "--" ]; do GR+=("$1"); shift; done
shift
export TMPDIR=/tmp
OUT=gpurun_out/pmcq_$TAG
mkdir -p "$OUT"
i=0
for g in "${GR[@]}"; do
  if rocprofv3 --pmc $g -d "$OUT/g$i" --output-format csv -- python3 bench.py --cpu-seconds 0 --steps 2 --warmup 1 --fresh-batches 1 --history-steps 0 "$@" > "$OUT/g$i.log" 2>&1; then
    echo "pmc group $i ($g) done"
  else
    echo "pmc group $i ($g) FAILED (skipped)"
  fi
  i=$((i+1))
done
python3 tools/pmc_summarise.py "$OUT" "$KERN" > "$OUT/pmc.json"
cat "$OUT/pmc.json"
