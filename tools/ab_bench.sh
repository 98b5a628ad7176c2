#!/bin/bash
# A/B of two builds of librqp_hip.so on the SAME GPU box (boxes differ by ~3 % in kernel time, so numbers of separate gpurun
# calls do not compare).  Build the two libraries here first:
#     git stash; make -C reluqp-py_amd/csrc; cp reluqp-py_amd/reluqp/lib/librqp_hip.so .ab/lib_old.so; git stash pop
#     make -C reluqp-py_amd/csrc;            cp reluqp-py_amd/reluqp/lib/librqp_hip.so .ab/lib_new.so
# (.ab/ is git-ignored but travels with the gpurun snapshot), then on the box:
#     gpurun -- 'bash tools/ab_bench.sh "" "--workload c4 --batch 65536"'
# Each argument is one bench.py argument string; old and new alternate, twice, 30 timed steps each.
L=reluqp-py_amd/reluqp/lib/librqp_hip.so
for args in "$@"; do
  for rep in 1 2; do
    for v in old new; do
      cp .ab/lib_$v.so $L
      printf "%s [%s] %s: " "$v" "$args" "$rep"
      timeout -k 10 150 python bench.py --cpu-seconds 0 --steps 20 --warmup 3 --fresh-batches 2 $args 2>&1 | tail -1 | grep -o "\"kernel_ms\": [0-9.]*" | tr '\n' ' '
      echo
    done
  done
done
cp .ab/lib_new.so $L
