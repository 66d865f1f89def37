# HBM-side traffic of the dominant conv kernel class: two SEPARATE counter passes (FETCH_SIZE, WRITE_SIZE), as
# MI355X_MICROARCH.md prescribes; --kernel-trace only.  Writes gpurun_out/pmc_{fetch,write}/ + a JSON summary.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_fetch -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-graph > /dev/null 2> $R/gpurun_out/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_write -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-graph > /dev/null 2> $R/gpurun_out/pmc_write.err
python3 $R/tools/pmc_traffic.py $R/gpurun_out/pmc_fetch $R/gpurun_out/pmc_write > $R/gpurun_out/pmc_traffic.json
cat $R/gpurun_out/pmc_traffic.json
