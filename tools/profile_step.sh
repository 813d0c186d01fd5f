#!/bin/bash
# Round profile of the headline command on the GPU box: rocprofv3 kernel-trace stats + the two PMC passes (FETCH_SIZE, WRITE_SIZE, each its
# own run with --kernel-trace only, MI355X_MICROARCH.md "HBM").  Usage (inside gpurun): bash tools/profile_step.sh <tag> [bench args]
# Writes gpurun_out/<tag>_kernel_stats.csv, gpurun_out/<tag>_pmc_traffic.json (copy both to profiles/ afterwards).
set -e
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
[ -z "$R" ] && R=/root/repo
ARGS="--steps 6 --warmup 2 --no-cpu-baseline --no-kernel-timer --no-forward-metric $*"
STEPS=8          # warm-up + timed
rm -rf /tmp/prof_kt /tmp/prof_f /tmp/prof_w
rocprofv3 --kernel-trace --stats -d /tmp/prof_kt -o kt -- python3 $R/bench.py $ARGS > $R/gpurun_out/${tag}_kt.log 2>&1
f=$(find /tmp/prof_kt -name "*kernel_stats.csv" | head -1)
if [ -n "$f" ]; then cp "$f" $R/gpurun_out/${tag}_kernel_stats.csv; else db=$(find /tmp/prof_kt -name "*.db" | head -1); python3 $R/tools/rocpd_stats.py "$db" > $R/gpurun_out/${tag}_kernel_stats.csv; fi
echo "kernel stats done"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/prof_f -o f -- python3 $R/bench.py $ARGS > $R/gpurun_out/${tag}_pmc_f.log 2>&1
echo "fetch pass done"
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/prof_w -o w -- python3 $R/bench.py $ARGS > $R/gpurun_out/${tag}_pmc_w.log 2>&1
echo "write pass done"
ff=$(find /tmp/prof_f -name "*counter_collection.csv" | head -1); [ -z "$ff" ] && ff=$(find /tmp/prof_f -name "*.db" | head -1)
fw=$(find /tmp/prof_w -name "*counter_collection.csv" | head -1); [ -z "$fw" ] && fw=$(find /tmp/prof_w -name "*.db" | head -1)
cd $R && python3 tools/pmc_traffic.py "$ff" "$fw" gpurun_out/${tag}_pmc_traffic.json $STEPS | head -30
