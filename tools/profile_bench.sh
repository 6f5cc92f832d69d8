#!/bin/bash
# GPU-box helper: rocprofv3 --kernel-trace --stats of the bench command + the bench JSON line itself.
# usage: tools/profile_bench.sh <tag>
TAG=${1:-r01}; OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
python $GRAFT_REPO_ROOT/bench.py --profile-ops $OUT/ops.txt > $OUT/bench.json 2> $OUT/bench.err
tail -1 $OUT/bench.json
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o b -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench_prof.json 2> $OUT/prof.err
find $OUT/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
head -12 $OUT/kernel_stats.csv
rm -rf $OUT/stats
