#!/bin/bash
# Round profile refresh on the GPU box (inside gpurun): kernel stats + PMC passes, single-stream stats, step timeline, then the
# bench log with matching traffic provenance.  bash tools/refresh_profiles.sh <tag>   (copy gpurun_out/<tag>_* into profiles/ afterwards)
tag=${1:-r02}
R=$GRAFT_REPO_ROOT
[ -z "$R" ] && R=/root/repo
bash $R/tools/profile_step.sh $tag > $R/gpurun_out/${tag}_profile.log 2>&1 || exit 1
cp $R/gpurun_out/${tag}_pmc_traffic.json $R/profiles/pmc_traffic.json
bash $R/tools/profile_single_stream.sh $tag > /dev/null 2>&1
bash $R/tools/step_timeline.sh ${tag}two > /dev/null 2>&1
cd $R && python bench.py > gpurun_out/${tag}_bench_c2.log 2>&1
tail -1 gpurun_out/${tag}_bench_c2.log | python tools/bench_summary.py /dev/stdin | head -1
