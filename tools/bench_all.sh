#!/bin/bash
# Un-profiled bench.py lines of every workload (GPU box).  Usage: bash tools/bench_all.sh <outdir> [cpu-seconds of the default run]
set -e
OUT=${1:-gpurun_out/bench_all}
CPU=${2:-0}
mkdir -p $OUT
python3 bench.py --cpu-seconds $CPU > $OUT/bench_resident2.json 2> $OUT/bench_resident2.err
python3 bench.py --cpu-seconds 0 --precision f64 > $OUT/bench_resident64.json 2> $OUT/bench_resident64.err
python3 bench.py --cpu-seconds 0 --workload c4 > $OUT/bench_wave.json 2> $OUT/bench_wave.err
python3 bench.py --cpu-seconds 0 --workload mpc > $OUT/bench_mfma.json 2> $OUT/bench_mfma.err
python3 bench.py --cpu-seconds 0 --workload mpc --tile bf16 > $OUT/bench_mfma16.json 2> $OUT/bench_mfma16.err
python3 bench.py --cpu-seconds 0 --workload mpc --mpc-form sparse > $OUT/bench_mfmal.json 2> $OUT/bench_mfmal.err
python3 bench.py --cpu-seconds 0 --workload mpc --precision f64 > $OUT/bench_mfmad.json 2> $OUT/bench_mfmad.err
python3 - <<PY
import json
for t in ["resident2", "resident64", "wave", "mfma", "mfma16", "mfmal", "mfmad"]:
    d = json.load(open("$OUT/bench_%s.json" % t))
    print("%-10s value %.0f  ms/step %.3f  kernel_ms %.3f  frac %.3f  setup_s %.4f  setup+solve %.0f  with_history %s" % (
        t, d["value"], d["ms_per_step"], d["roofline"]["kernel_ms"], d["roofline"]["frac"], d["setup_s"], d["setup_plus_solve_qps"],
        d.get("value_with_history")))
PY
