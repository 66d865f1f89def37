cd $GRAFT_REPO_ROOT
export DAHITRA_HIP_LIB=$GRAFT_REPO_ROOT/build/exp/lib_up4_dbg.so
O=$GRAFT_REPO_ROOT/gpurun_out/r06a_up4pmc; mkdir -p $O
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT --kernel-trace --output-format csv -d $O/p1 -- python3 $R/tools/up4_bench.py > /dev/null 2> $O/p1.err
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/p2 -- python3 $R/tools/up4_bench.py > /dev/null 2> $O/p2.err
python3 - $O <<'PY'
import collections, csv, glob, sys
for which in ("p1", "p2"):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for f in glob.glob("%s/%s/*/*counter_collection.csv" % (sys.argv[1], which)):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "wreg32" in k or "absdiff_up4_fwd" in k:
                key = k.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
                acc[key][r["Counter_Name"]] += float(r["Counter_Value"]); n[(key, r["Counter_Name"])] += 1
    for k, v in sorted(acc.items()):
        print(which, k, {c: "%.3g" % (x / n[(k, c)]) for c, x in v.items()})
PY
rm -rf $O/p1 $O/p2
