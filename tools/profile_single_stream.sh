#!/bin/bash
# rocprofv3 kernel-trace summary of the headline step with the weight gradients on the MAIN stream (every kernel alone on the whole
# chip: the counterpart of bench.py's `roofline` timing).  Inside gpurun: bash tools/profile_single_stream.sh <tag>
tag=${1:-r02}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
[ -z "$R" ] && R=/root/repo
export MRISR_WGRAD_STREAM=0
rm -rf /tmp/prof_ss
rocprofv3 --kernel-trace --stats -d /tmp/prof_ss -o ss -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-kernel-timer --no-forward-metric > $R/gpurun_out/${tag}_ss.log 2>&1
f=$(find /tmp/prof_ss -name "*kernel_stats.csv" | head -1)
if [ -n "$f" ]; then cp "$f" $R/gpurun_out/${tag}_kernel_stats_single_stream.csv; else db=$(find /tmp/prof_ss -name "*.db" | head -1); python3 $R/tools/rocpd_stats.py "$db" > $R/gpurun_out/${tag}_kernel_stats_single_stream.csv; fi
head -5 $R/gpurun_out/${tag}_kernel_stats_single_stream.csv
