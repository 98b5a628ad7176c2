#!/bin/bash
# PMC passes of the setup kernels of the default bench.py workload (GPU box): MFMA busy / activity of k_gram_mfma and VALU activity of
# k_factor_reg2, one counter group per rocprofv3 run, no trace domains.  Writes gpurun_out/pmc_setup/{gram,factor}.json
export TMPDIR=/tmp
OUT=gpurun_out/pmc_setup
mkdir -p $OUT
i=0
for g in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA" "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_LDS" "FETCH_SIZE" "WRITE_SIZE"; do
  if rocprofv3 --pmc $g -d "$OUT/g$i" --output-format csv -- python3 bench.py --cpu-seconds 0 --steps 1 --warmup 1 --fresh-batches 2 --history-steps 0 > "$OUT/g$i.log" 2>&1; then echo "group $i ok"; else echo "group $i FAILED"; fi
  i=$((i+1))
done
python3 tools/pmc_summarise.py $OUT k_gram_mfma > $OUT/gram.json
python3 tools/pmc_summarise.py $OUT k_factor_reg2 > $OUT/factor.json
python3 tools/pmc_summarise.py $OUT k_pack_res2_ah > $OUT/pack_ah.json
cat $OUT/gram.json
