#!/bin/bash
# kernel-trace stats of one bench.py command (GPU box): tools/kstats.sh TAG [bench args...] -> gpurun_out/kstats_TAG.csv
# (never ends in a command that could read stdin: a missing csv is reported, not waited for)
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/kst_$TAG -o st -- python3 $R/bench.py --cpu-seconds 0 --steps 8 --history-steps 0 "$@" > $R/gpurun_out/kst_$TAG.json 2> $R/gpurun_out/kst_$TAG.err
cd $R
F=$(find gpurun_out/kst_$TAG -name "*kernel_stats.csv" | head -1)
if [ -n "$F" ] && [ -f "$F" ]; then cp "$F" gpurun_out/kstats_$TAG.csv; cut -c1-160 gpurun_out/kstats_$TAG.csv | sed -n 1,9p; else echo "no kernel_stats.csv under gpurun_out/kst_$TAG"; fi
