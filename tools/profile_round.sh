# usage (on the GPU box): bash tools/profile_round.sh <tag>
# Every profile the bench line quotes, in one go, written under gpurun_out/<tag>/ with the names they get in profiles/:
#   kernel-trace stats of the default bench command, the four --pmc passes (FETCH_SIZE, WRITE_SIZE, MFMA busy, LDS conflicts;
#   each its own run with --kernel-trace only, as MI355X_MICROARCH.md prescribes) and their JSON summaries.
tag=$1
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$tag
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-parity-mode --no-class-replay > $O/${tag}_prof_bench.json 2> $O/stats.err
cp $(ls $O/stats/*/*kernel_stats.csv | head -1) $O/${tag}_bench_bf16_b32_kernel_stats.csv
B="python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-parity-mode --no-class-replay --no-graph"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -- $B > /dev/null 2> $O/fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -- $B > /dev/null 2> $O/write.err
rocprofv3 --pmc SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d $O/util -- $B > /dev/null 2> $O/util.err
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $O/lds -- $B > /dev/null 2> $O/lds.err
python3 $R/tools/pmc_step.py $O/fetch $O/write $O/util $O/lds $O/$tag | head -30
rm -rf $O/fetch $O/write $O/util $O/lds $O/stats
ls -la $O
