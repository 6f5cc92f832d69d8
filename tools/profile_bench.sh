#!/bin/bash
# GPU-box helper: the evidence bundle of a round.  usage: tools/profile_bench.sh <tag>   (writes gpurun_out/prof_<tag>/)
#   bench.json            : the default bench line (with cpu_baseline) + per-op HIP-event table ops.txt
#   kernel_stats.csv      : rocprofv3 --kernel-trace --stats of the same command (shorter run, concurrent streams)
#   seq_trace_by_op.md    : sequential run (one trunk context: nothing overlaps the batched per-t sequence) traced and mapped back to the plan ops = true in-sequence kernel times
TAG=${1:-r02}; OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --profile-ops $OUT/ops.txt > $OUT/bench.json 2> $OUT/bench.err
tail -c 600 $OUT/bench.json; echo
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o b -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline > $OUT/bench_prof.json 2> $OUT/prof.err
find $OUT/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
head -8 $OUT/kernel_stats.csv | cut -c1-160
rm -rf $OUT/stats
DEMFI_NTRUNK=1 rocprofv3 --kernel-trace --output-format csv -d $OUT/seq -o s -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --profile-ops $OUT/ops_seq.txt > $OUT/bench_seq.json 2> $OUT/seq.err
T=$(find $OUT/seq -name "*kernel_trace.csv" | head -1)
python $GRAFT_REPO_ROOT/tools/trace_by_op.py $T $OUT/ops_seq.txt $OUT/seq_trace_by_op.md 1 $OUT/seq_trace_roofline.json 2>&1 | tail -3
rm -rf $OUT/seq
head -5 $OUT/seq_trace_by_op.md
# fp16-vs-oracle margins recorded by tests/test_gpu_configs.py (run `pytest tests/test_gpu_configs.py -m gpu` in the same gpurun call)
[ -f $GRAFT_REPO_ROOT/gpurun_out/fp16_margins.json ] && cp $GRAFT_REPO_ROOT/gpurun_out/fp16_margins.json $OUT/fp16_margins.json
