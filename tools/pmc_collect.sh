#!/bin/bash
# PMC passes of ONE bench.py command, one counter group per rocprofv3 run (MI355X_MICROARCH.md: separate --pmc passes; never
# combined with trace domains).  Usage on the GPU box:
#   bash tools/pmc_collect.sh <tag> <kernel-name-substring> -- <bench.py args...>
# Writes gpurun_out/pmc_<tag>/<group>/ and the per-launch means of the named kernel to gpurun_out/pmc_<tag>/pmc.json
# (copy that file into profiles/<round>_<kernel>/pmc.json).
set -e
TAG=$1; KERN=$2; shift 3
export TMPDIR=/tmp
OUT=gpurun_out/pmc_$TAG
mkdir -p "$OUT"
GROUPS_=("FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS"
         "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES"
         "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_MFMA")
i=0
for g in "${GROUPS_[@]}"; do
  # (one batch, no with-history leg: the counters of a kernel are averaged over its launches; a group this build of rocprofv3
  #  does not know is skipped, not fatal)
  if rocprofv3 --pmc $g -d "$OUT/g$i" --output-format csv -- python3 bench.py --cpu-seconds 0 --steps 2 --warmup 1 --fresh-batches 1 --history-steps 0 "$@" > "$OUT/g$i.log" 2>&1; then
    echo "pmc group $i ($g) done"
  else
    echo "pmc group $i ($g) FAILED (skipped)"
  fi
  i=$((i+1))
done
python3 tools/pmc_summarise.py "$OUT" "$KERN" > "$OUT/pmc.json"
cat "$OUT/pmc.json"
