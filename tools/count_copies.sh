R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for st in 4 14; do
rm -rf /tmp/cc; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/cc -- python3 $R/bench.py --net newUNetTrans --steps $st --warmup 2 --no-cpu-baseline --no-parity-mode > /dev/null 2>&1
echo "steps=$st"; grep -i "copyBuffer\|FillFunctor<float>\|adamw_tick\|fillBuffer" $(ls /tmp/cc/*/*kernel_stats.csv | head -1) | cut -d, -f1,2 | cut -c1-90
done
