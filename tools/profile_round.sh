#!/bin/bash
# Round profile set (GPU box): kernel-trace stats + PMC passes of the three bench workloads.  Outputs under gpurun_out/prof_$1/.
# Usage: bash tools/profile_round.sh r2
set -e
R=${1:-r3}
export TMPDIR=/tmp
OUT=gpurun_out/prof_$R
mkdir -p $OUT
run_stats () {  # tag, bench args...
  local tag=$1; shift
  rocprofv3 --kernel-trace --stats -d $OUT/stats_$tag --output-format csv -- python3 bench.py --cpu-seconds 0 --steps 5 --warmup 2 "$@" > $OUT/bench_$tag.json 2> $OUT/bench_$tag.err
  find $OUT/stats_$tag -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats_$tag.csv \;
  echo "stats $tag done"
}
run_stats resident2
run_stats mfma --workload mpc
run_stats mfma16 --workload mpc --tile bf16
run_stats mfmal --workload mpc --mpc-form sparse
run_stats mfmad --workload mpc --precision f64
run_stats wave --workload c4
run_stats resident64 --precision f64
bash tools/pmc_collect.sh ${R}_resident2 k_admm_res2 -- > $OUT/pmc_resident2.log 2>&1 && cp gpurun_out/pmc_${R}_resident2/pmc.json $OUT/pmc_resident2.json
bash tools/pmc_collect.sh ${R}_mfma "k_admm_mfma<" -- --workload mpc > $OUT/pmc_mfma.log 2>&1 && cp gpurun_out/pmc_${R}_mfma/pmc.json $OUT/pmc_mfma.json
bash tools/pmc_collect.sh ${R}_mfma16 k_admm_mfma16 -- --workload mpc --tile bf16 > $OUT/pmc_mfma16.log 2>&1 && cp gpurun_out/pmc_${R}_mfma16/pmc.json $OUT/pmc_mfma16.json
bash tools/pmc_collect.sh ${R}_mfmal k_admm_mfmal -- --workload mpc --mpc-form sparse > $OUT/pmc_mfmal.log 2>&1 && cp gpurun_out/pmc_${R}_mfmal/pmc.json $OUT/pmc_mfmal.json
bash tools/pmc_collect.sh ${R}_mfmad k_admm_mfmad -- --workload mpc --precision f64 > $OUT/pmc_mfmad.log 2>&1 && cp gpurun_out/pmc_${R}_mfmad/pmc.json $OUT/pmc_mfmad.json
bash tools/pmc_collect.sh ${R}_wave k_admm_wave -- --workload c4 > $OUT/pmc_wave.log 2>&1 && cp gpurun_out/pmc_${R}_wave/pmc.json $OUT/pmc_wave.json
bash tools/pmc_collect.sh ${R}_resident64 k_admm_res64 -- --precision f64 > $OUT/pmc_resident64.log 2>&1 && cp gpurun_out/pmc_${R}_resident64/pmc.json $OUT/pmc_resident64.json
ls -la $OUT
