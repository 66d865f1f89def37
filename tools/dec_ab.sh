# usage (GPU box): bash tools/dec_ab.sh <prev-lib.so> <tag>
# fused decoder layer, previous build vs the tree's: tools/dec_bench.py timings (the new outputs checked against the previous
# build's), then one --pmc pass each for the LDS bank-conflict share of every decoder kernel (gpurun_out/<tag>/)
prev=$1; tag=${2:-dec_ab}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$tag
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export DAHITRA_HIP_LIB=$prev
python3 $R/tools/dec_bench.py --save $O/ref.pt > $O/bench_prev.txt 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $O/lds_prev -- python3 $R/tools/dec_bench.py > /dev/null 2> $O/lds_prev.err
unset DAHITRA_HIP_LIB
python3 $R/tools/dec_bench.py --check $O/ref.pt > $O/bench_new.txt 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $O/lds_new -- python3 $R/tools/dec_bench.py > /dev/null 2> $O/lds_new.err
python3 - $O <<'PY'
import collections, csv, glob, sys
for which in ("prev", "new"):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    for f in glob.glob("%s/lds_%s/*/*counter_collection.csv" % (sys.argv[1], which)):
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"]
            if "dec_" in n:
                acc[n.split("(")[0].replace("(anonymous namespace)::", "").replace("void ", "")][r["Counter_Name"]] += float(r["Counter_Value"])
    for k, v in sorted(acc.items()):
        print("%-5s %-28s LDS bank conflict / active = %.3f   (active %.3g)" % (which, k, v["SQ_LDS_BANK_CONFLICT"] / max(v["SQ_LDS_IDX_ACTIVE"], 1), v["SQ_LDS_IDX_ACTIVE"]))
PY
rm -rf $O/lds_prev $O/lds_new $O/ref.pt
echo "--- previous build"; cat $O/bench_prev.txt; echo "--- this tree"; cat $O/bench_new.txt
