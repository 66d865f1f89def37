# usage (on the GPU box): bash tools/profile_round.sh <tag> [net] [label extra bench args...]
#   e.g.  bash tools/profile_round.sh r05k base_transformer_pos_s4_resnet50 r50_1024 --img 1024 --batch 8
#         bash tools/profile_round.sh r05k base_transformer_pos_s4 s4_bf16x3 --dtype bf16x3
# Every profile the bench line quotes, in one go, written under gpurun_out/<tag>/ with the names they get in profiles/
# (<tag>_... for the headline net, <tag>_<net>_... for any other: bench.py's _profile() looks them up by profiles/CURRENT):
#   kernel-trace stats of the bench command, the four --pmc passes (FETCH_SIZE, WRITE_SIZE, MFMA busy, LDS conflicts;
#   each its own run with --kernel-trace only, as MI355X_MICROARCH.md prescribes), their JSON summaries, and the kernel
#   sequence of one graph-replayed step (launch count, kernel time, gaps).
tag=$1
net=${2:-base_transformer_pos_s4}
label=$3
if [ $# -ge 3 ]; then shift 3; else shift $#; fi
EXTRA="$@"
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$tag
mkdir -p $O
pre=${tag}
stats=${tag}_bench_bf16_b32
if [ "$net" != "base_transformer_pos_s4" ]; then pre=${tag}_${net}; stats=${tag}_bench_${net}; fi
if [ -n "$label" ]; then pre=${tag}_${label}; stats=${tag}_bench_${label}; fi
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --net $net $EXTRA --steps 6 --warmup 2 --no-cpu-baseline --no-parity-mode --no-class-replay --no-secondary --no-ddp-rehearsal > $O/${pre}_prof_bench.json 2> $O/stats.err
cp $(ls $O/stats/*/*kernel_stats.csv | head -1) $O/${stats}_kernel_stats.csv
python3 - $O/stats $O/${pre}_graph_step.txt <<'PY'
import csv, glob, sys, re
f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "adamw_tick" in r["Kernel_Name"] or "adamw_kernel" in r["Kernel_Name"] or "adamw_xbd_tick" in r["Kernel_Name"]]
span = lambda k: int(rows[idx[k + 1]]["End_Timestamp"]) - int(rows[idx[k] + 1]["Start_Timestamp"])
k = min(range(len(idx) - 1), key=span)          # the fastest step = a replay of the recorded graph
lo, hi = idx[k] + 1, idx[k + 1] + 1
prev = int(rows[lo]["Start_Timestamp"])
with open(sys.argv[2], "w") as o:
    tot = gaps = small = copies = 0
    for i in range(lo, hi):
        r = rows[i]
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        n = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])[:110]
        o.write("%4d %8.1f %7.1f  %s\n" % (i - lo, (e - s) / 1e3, (s - prev) / 1e3, n))
        tot += e - s; gaps += max(s - prev, 0); prev = e
        small += (e - s) < 8000
        copies += "copyBuffer" in n or "fillBuffer" in n
    o.write("graph-replayed step: kernels %d (of them < 8 us: %d; runtime copy / fill kernels: %d), kernel time %.1f us, gaps %.1f us, span %.1f us\n"
            % (hi - lo, small, copies, tot / 1e3, gaps / 1e3, (int(rows[hi - 1]["End_Timestamp"]) - int(rows[lo]["Start_Timestamp"])) / 1e3))
print(open(sys.argv[2]).read().splitlines()[-1])
PY
B="python3 $R/bench.py --net $net $EXTRA --steps 3 --warmup 1 --no-cpu-baseline --no-parity-mode --no-class-replay --no-secondary --no-ddp-rehearsal --no-graph"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -- $B > /dev/null 2> $O/fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -- $B > /dev/null 2> $O/write.err
rocprofv3 --pmc SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d $O/util -- $B > /dev/null 2> $O/util.err
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $O/lds -- $B > /dev/null 2> $O/lds.err
python3 $R/tools/pmc_step.py $O/fetch $O/write $O/util $O/lds $O/$pre | tail -4
rm -rf $O/fetch $O/write $O/util $O/lds $O/stats
ls $O
