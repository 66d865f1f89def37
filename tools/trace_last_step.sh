# usage (GPU box): bash tools/trace_last_step.sh <tag> [bench.py args]  -- kernel sequence of the fastest (graph-replayed) step of a bench run
# (rocprofv3 --kernel-trace), written to gpurun_out/<tag>_last_step.txt: index, duration us, gap to the previous kernel, name
tag=$1; shift
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$tag
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/tr -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-parity-mode --no-class-replay "$@" > /dev/null 2> $O/tr.err
python3 - $O/tr $R/gpurun_out/${tag}_last_step.txt <<'PY'
import csv, glob, sys, re
f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "adamw_tick" in r["Kernel_Name"] or "adamw_kernel" in r["Kernel_Name"] or "adamw_xbd_tick" in r["Kernel_Name"]]
# the step with the SHORTEST span between two optimizer launches: a replay of the recorded graph (the run's last steps are the
# eager profiling steps of bench.py, whose launches come from Python one by one)
span = lambda k: int(rows[idx[k + 1]]["End_Timestamp"]) - int(rows[idx[k] + 1]["Start_Timestamp"])
k = min(range(len(idx) - 1), key=span)
lo, hi = idx[k] + 1, idx[k + 1] + 1
prev = int(rows[lo - 1]["End_Timestamp"])
with open(sys.argv[2], "w") as o:
    tot = gaps = 0
    for i in range(lo, hi):
        r = rows[i]
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        n = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])[:110]
        o.write("%4d %8.1f %7.1f  %s\n" % (i - lo, (e - s) / 1e3, (s - prev) / 1e3, n))
        tot += e - s; gaps += max(s - prev, 0); prev = e
    o.write("kernels %d, kernel time %.1f us, gaps %.1f us, span %.1f us\n" % (hi - lo, tot / 1e3, gaps / 1e3, (int(rows[hi-1]["End_Timestamp"]) - int(rows[lo]["Start_Timestamp"])) / 1e3))
PY
rm -rf $O
