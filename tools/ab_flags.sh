#!/bin/bash
# A/B of two bench.py argument strings on the SAME GPU box with the SAME library (boxes differ by ~3 %): the arguments
# alternate, REPS times each.  Usage: gpurun -- 'bash tools/ab_flags.sh 3 "" "--full-ladder"'
REPS=$1; shift
for rep in $(seq 1 $REPS); do
  for args in "$@"; do
    printf "[%s] %s: " "$args" "$rep"
    timeout -k 10 150 python bench.py --cpu-seconds 0 --steps 30 --warmup 3 $args 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('kernel_ms %.3f  ms/step %.3f  hist_kernel_ms %.3f  setup_ms %.2f' % (r['kernel_ms'], d['ms_per_step'], d.get('with_history',{}).get('kernel_ms',0), d['setup_s']*1e3))"
  done
done
