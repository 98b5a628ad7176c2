#!/bin/bash
# bench.py JSON lines of the round WITHOUT a profiler -> gpurun_out/lines_$1/*.json   (bash tools/bench_lines.sh r2)
R=${1:-r2}; OUT=gpurun_out/lines_$R; mkdir -p $OUT
python bench.py > $OUT/resident2_default.json 2> $OUT/err.log
python bench.py --cpu-seconds 0 --workload mpc > $OUT/mfma_mpc.json 2>> $OUT/err.log
python bench.py --cpu-seconds 0 --workload c4 > $OUT/wave_c4_8192.json 2>> $OUT/err.log
python bench.py --cpu-seconds 0 --workload c4 --batch 65536 > $OUT/wave_c4_65536.json 2>> $OUT/err.log
python bench.py --cpu-seconds 0 --precision f64 > $OUT/resident64_f64.json 2>> $OUT/err.log
python bench.py --cpu-seconds 0 --tile f16 > $OUT/resident2_tile_f16.json 2>> $OUT/err.log
python bench.py --cpu-seconds 0 --batch 1024 > $OUT/resident2_c2_b1024.json 2>> $OUT/err.log
for f in $OUT/*.json; do python - "$f" <<'PY'
import json,sys
l=[x for x in open(sys.argv[1]) if x.startswith('{')]
if l:
    d=json.loads(l[-1]); r=d['roofline']
    print(sys.argv[1].split('/')[-1], '%.4g QP/s'%d['value'], '%.3f ms/step'%d['ms_per_step'], r['bound'], 'frac %.3f'%r['frac'], 'kernel %.3f ms'%r['kernel_ms'], 'first %.3f'%r.get('kernel_ms_first_launch',0), 'setup %.1f ms'%(d['setup_s']*1e3))
PY
done
