# usage: bash tools/prof_bench.sh <tag> [bench args...]   -> gpurun_out/prof_<tag>/ kernel stats csv
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_$tag -- python3 $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-class-replay --no-secondary --no-ddp-rehearsal "$@" > $GRAFT_REPO_ROOT/gpurun_out/prof_$tag.json 2> $GRAFT_REPO_ROOT/gpurun_out/prof_$tag.err
f=$(ls $GRAFT_REPO_ROOT/gpurun_out/prof_$tag/*/*kernel_stats.csv | head -1)
cp $f $GRAFT_REPO_ROOT/gpurun_out/prof_${tag}_kernel_stats.csv
head -40 $f | cut -c1-200
