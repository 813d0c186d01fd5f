#!/bin/bash
# One-step kernel timeline on the GPU box (inside gpurun): bash tools/step_timeline.sh <tag>  -> gpurun_out/<tag>_timeline.txt
# (environment switches such as MRISR_WGRAD_STREAM pass through to bench.py)
tag=${1:-step}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
[ -z "$R" ] && R=/root/repo
rm -rf /tmp/prof_tl_$tag
rocprofv3 --kernel-trace -d /tmp/prof_tl_$tag -o tl -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-timer --no-forward-metric $BENCH_ARGS > $R/gpurun_out/${tag}_tl.log 2>&1
db=$(find /tmp/prof_tl_$tag -name "*.db" | head -1)
python3 $R/tools/step_timeline.py "$db" --full > $R/gpurun_out/${tag}_timeline.txt 2>&1
tail -4 $R/gpurun_out/${tag}_timeline.txt
