# usage (GPU box): bash tools/prof_unet.sh <tag>  -- kernel-trace stats of the newUNetTrans bench step + the per-launch
# durations of the fused decoder kernels by grid size (gpurun_out/<tag>/)
tag=${1:-unet}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$tag
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --net newUNetTrans --steps 6 --warmup 2 --no-cpu-baseline --no-parity-mode --no-class-replay > $O/prof_bench.json 2> $O/stats.err
cp $(ls $O/stats/*/*kernel_stats.csv | head -1) $O/${tag}_bench_newUNetTrans_kernel_stats.csv
python3 - $O/stats $O/${tag}_decoder_launches.txt <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    if "dec_bwd" in n or "dec_fwd" in n:          # dec_{fwd,bwd}_kernel and the staged levels' dec_{fwd,bwd}_multi_kernel, keyed by name
        key = (n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0], int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1))
        acc[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
with open(sys.argv[2], "w") as o:
    for k in sorted(acc):
        v = sorted(acc[k])
        o.write("%-44s grid %5d: %4d launches, median %.1f us, min %.1f\n" % (k[0], k[1], len(v), v[len(v) // 2], v[0]))
print(open(sys.argv[2]).read())
PY
rm -rf $O/stats
