"""Summarise the rocprofv3 --pmc passes of one bench.py run (tools/profile_round.sh) into the JSON files under profiles/.

usage: pmc_step.py <fetch_dir> <write_dir> <util_dir> <lds_dir> <out_prefix>

  <out_prefix>_pmc_traffic_conv3x3.json   per-launch HBM-side traffic of the 3x3 stride-1 MFMA conv kernels (the constant
                                          bench.py's roofline.traffic is read from)
  <out_prefix>_pmc_step_traffic.json      FETCH/WRITE of every kernel per training step and the step total
  <out_prefix>_pmc_mfma_util.json         SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CU_CYCLES and LDS bank-conflict share per kernel

FETCH_SIZE / WRITE_SIZE are reported in KB; FETCH_SIZE is doubled (gfx950 counts 128-byte read requests at 64 B,
MI355X_MICROARCH.md section HBM).  One training step = one adamw_tick_kernel (graph-capturable optimizer) or adamw_kernel launch, which is how launches are turned
into per-step figures."""
import collections
import csv
import glob
import json
import re
import sys


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    m = re.match(r"([A-Za-z0-9_:]+(<[^(]*>)?)", name)
    return m.group(1) if m else name


def load(d):
    """{kernel: {counter: [launches, sum]}}"""
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
    for f in glob.glob(d + "/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            e = acc[short(r["Kernel_Name"])][r["Counter_Name"]]
            e[0] += 1
            e[1] += float(r["Counter_Value"])
    return acc


def steps_of(acc):
    for name in ("adamw_tick_kernel", "adamw_xbd_tick_kernel", "adamw_kernel"):          # capturable / plain optimizer step: one launch per step
        for k, v in acc.items():
            if k.startswith(name):
                return next(iter(v.values()))[0]
    raise SystemExit("no adamw launches in the pass: cannot count steps")


fetch, write, util, lds, prefix = load(sys.argv[1]), load(sys.argv[2]), load(sys.argv[3]), load(sys.argv[4]), sys.argv[5]

# ---- per-launch traffic of the dominant conv class ----
conv = {}
for k in sorted(fetch):
    if not (k.startswith("conv_mfma_kernel<bf16, 3, 1, 64") or k.startswith("conv3x3_wreg_kernel")):
        continue
    n, kb = fetch[k]["FETCH_SIZE"]
    wn, wkb = write.get(k, {}).get("WRITE_SIZE", [0, 0.0])
    conv[k] = {"launches": n, "fetch_MB_per_launch_raw": kb / n / 1024.0, "fetch_MB_per_launch_corrected_x2": 2.0 * kb / n / 1024.0,
               "write_MB_per_launch": (wkb / wn / 1024.0) if wn else None}
json.dump(conv, open(prefix + "_pmc_traffic_conv3x3.json", "w"), indent=1)

# ---- whole step ----
sf, sw = steps_of(fetch), steps_of(write)
rows, tf, tw = {}, 0.0, 0.0
for k in sorted(set(fetch) | set(write)):
    n, kb = fetch.get(k, {}).get("FETCH_SIZE", [0, 0.0])
    wn, wkb = write.get(k, {}).get("WRITE_SIZE", [0, 0.0])
    f_mb, w_mb = 2.0 * kb / 1024.0 / sf, wkb / 1024.0 / sw
    tf, tw = tf + f_mb, tw + w_mb
    if f_mb + w_mb >= 0.5:
        rows[k] = {"launches_per_step": round(n / sf, 2), "fetch_MB_per_step_x2": round(f_mb, 1), "write_MB_per_step": round(w_mb, 1)}
rows = dict(sorted(rows.items(), key=lambda kv: -(kv[1]["fetch_MB_per_step_x2"] + kv[1]["write_MB_per_step"])))
json.dump({"steps_in_fetch_pass": sf, "steps_in_write_pass": sw, "fetch_MB_per_step_x2": round(tf, 1), "write_MB_per_step": round(tw, 1),
           "total_MB_per_step": round(tf + tw, 1), "note": "eager launches (--no-graph), one step = one adamw_tick_kernel; FETCH_SIZE x2",
           "kernels": rows}, open(prefix + "_pmc_step_traffic.json", "w"), indent=1)

# ---- MFMA / LDS utilisation ----
out = {}
for k in sorted(util):
    u = util[k]
    busy = u.get("SQ_BUSY_CU_CYCLES", [0, 0.0])[1]
    mf = u.get("SQ_VALU_MFMA_BUSY_CYCLES", [0, 0.0])[1]
    if busy <= 0 or mf <= 0:
        continue
    e = {"launches": u["SQ_BUSY_CU_CYCLES"][0], "mfma_busy_over_cu_busy": round(mf / busy, 4), "mfma_util": round(mf / busy / 4, 4)}
    ld = lds.get(k, {})
    act, conf = ld.get("SQ_LDS_IDX_ACTIVE", [0, 0.0])[1], ld.get("SQ_LDS_BANK_CONFLICT", [0, 0.0])[1]
    if act > 0:
        e["lds_bank_conflict_over_active"] = round(conf / act, 4)
    out[k] = e
json.dump({"note": "sums over all launches of the pass; SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CU_CYCLES as read (both per-SE "
                   "aggregates of rocprofv3; the ratio is what is comparable between kernels), SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE; "
                   "mfma_util = that ratio / 4 (the MFMA counter sums the four SIMDs of a CU, the CU-busy counter counts the CU once)",
           "kernels": out}, open(prefix + "_pmc_mfma_util.json", "w"), indent=1)
print(json.dumps({"conv": conv, "step_total_MB": round(tf + tw, 1), "steps": sf}, indent=1))
