"""Condense the rocprofv3 passes of tools/profile_r4.sh into one JSON: per workload and kernel, per-launch averages of every
counter plus the derived figures bench.py attaches to its `roofline` record.  usage: pmc_to_json.py <prof dir>"""
import csv, glob, hashlib, json, os, re, sys, collections

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
prof = sys.argv[1]
WORK = {"head": "1920x1080 spp=64 depth=4 region=256", "c5": "3840x2160 spp=1024 depth=8 region=1024",
        "c4": "3840x2160 spp=256 depth=8 region=256", "c5t": "3840x2160 spp=1024 depth=8 region=1024 pose=terrain",
        "ref": "1024x1024 spp=1 depth=2 region=256"}
SIMDS = 256 * 4


def source_sha16():
    h = hashlib.sha256()
    d = os.path.join(ROOT, "raytrace_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".hpp")) and name != "rt_api.hip":      # device code only: rt_api.hip is host-side
            h.update(open(os.path.join(d, name), "rb").read())
    h.update(open(os.path.join(ROOT, "include", "rt_math.h"), "rb").read())
    return h.hexdigest()[:16]


def short(name):
    n = name.split("(")[0].replace("void ", "").replace("rtd::", "")
    return n.split("<")[0], n


def counters(tagdir):
    agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
    for f in glob.glob(tagdir + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            a = agg[k][r["Counter_Name"]]
            a[0] += float(r["Counter_Value"]); a[1] += 1
    return agg


def stats(tagdir):
    out = {}
    for f in glob.glob(tagdir + "/**/*kernel_stats.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            out[short(r["Name"])] = {"calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"]), "min_ns": float(r["MinNs"]), "max_ns": float(r["MaxNs"])}
    return out


res = {"kernel_source_sha16": source_sha16(), "tool": "tools/profile_r4.sh (rocprofv3, one --pmc set per pass)", "workloads": {}}
# issue cost of the step loop's instruction mix (tools/isa_hist.py), if it was taken on these very sources
CPI = None
try:
    _h = json.load(open(os.path.join(ROOT, "profiles", "r4_step_loop_isa_hist.json")))
    if _h.get("kernel_source_sha16") == res["kernel_source_sha16"]:
        CPI = float(_h["cycles_per_valu_inst"])
except Exception:
    pass
for tag, wname in WORK.items():
    per = collections.defaultdict(dict)
    st = stats(os.path.join(prof, tag + "_stats"))
    for (k, full), v in st.items():
        if re.search(r"k_paths<true|k_persist<\d+, (true|false), true,|k_primary2<\d+, (true|false), true>|k_frame<\d+, (true|false), true>|k_accumulate_paths<false>", full):
            continue   # counting builds run once outside the timed region
        if k in ("k_paths", "k_persist", "k_frame", "k_primary2", "k_accumulate_paths"):
            per[k]["instantiation"] = full
            per[k]["avg_launch_ms"] = round(v["avg_ns"] / 1e6, 4); per[k]["min_launch_ms"] = round(v["min_ns"] / 1e6, 4); per[k]["launches_profiled"] = v["calls"]
    for sub in ("fetch", "write", "tcc", "sq1", "sq2", "sq3"):
        agg = counters(os.path.join(prof, "%s_%s" % (tag, sub)))
        for (k, full), cs in agg.items():
            if k not in per or per[k].get("instantiation") != full:
                continue
            raw = per[k].setdefault("raw", {})
            for c, (v, n) in cs.items():
                raw[c] = {"sum": v, "dispatches": n, "per_launch": v / n}
    # the bench line of the kernel-trace pass: how the profiled context sized its launches (bench.py compares)
    spl = None
    try:
        for l in open(os.path.join(prof, tag + "_stats.log")):
            if l.startswith("{"):
                spl = json.loads(l)["config"].get("samples_per_launch")
    except Exception:
        pass
    for k, r in per.items():
        if spl is not None:
            r["samples_per_launch"] = spl
        raw = r.get("raw", {})
        g = lambda c: raw[c]["per_launch"] if c in raw else None
        if g("FETCH_SIZE") is not None and g("WRITE_SIZE") is not None:
            # MI355X_MICROARCH.md: FETCH_SIZE counts 64 B per 128-B fabric read on gfx950 -> x2; WRITE_SIZE exact; both in KiB
            r["hbm_bytes_per_launch"] = int(g("FETCH_SIZE") * 2 * 1024 + g("WRITE_SIZE") * 1024)
            r["fetch_bytes_per_launch"] = int(g("FETCH_SIZE") * 2 * 1024); r["write_bytes_per_launch"] = int(g("WRITE_SIZE") * 1024)
        if g("TCC_HIT_sum") is not None:
            r["l2_hit_rate"] = round(g("TCC_HIT_sum") / (g("TCC_HIT_sum") + g("TCC_MISS_sum")), 4)
        if g("SQ_INSTS_VALU") is not None and g("SQ_BUSY_CYCLES") is not None:
            cycles = g("SQ_BUSY_CYCLES") / 32.0          # summed over 32 shader engines' SQs -> cycles of the launch
            wc = g("SQ_WAVE_CYCLES")
            r["valu"] = {
                "insts_per_launch": int(g("SQ_INSTS_VALU")),
                "salu_insts_per_launch": int(g("SQ_INSTS_SALU")) if g("SQ_INSTS_SALU") else None,
                "launch_cycles": int(cycles),
                # fraction of SIMD issue time at the two measured issue costs (tools/ubench/valu_rate.hip): 2 cycles per
                # wave64 instruction for fma/mul/add/and/or/lshr/mov, 3.4 for select/compare/convert/floor/bfi/min3/packed
                "pipe_busy_if_all_2cyc": round(g("SQ_INSTS_VALU") * 2.0 / (SIMDS * cycles), 3),
                "pipe_busy_if_all_3.4cyc": round(g("SQ_INSTS_VALU") * 3.4 / (SIMDS * cycles), 3),
                # ... and at the measured cost of the step loop's own instruction mix (profiles/r4_step_loop_isa_hist.json:
                # per-opcode counts of the step group x the micro-benchmark's issue cost), k_paths only
                "pipe_busy_weighted": round(g("SQ_INSTS_VALU") * CPI / (SIMDS * cycles), 3) if (CPI and k == "k_paths") else None,
                "cycles_per_valu_inst": CPI if k == "k_paths" else None,
                "exec_lane_fill": round(g("SQ_THREAD_CYCLES_VALU") / (64.0 * g("SQ_ACTIVE_INST_VALU")), 3) if g("SQ_THREAD_CYCLES_VALU") else None,
                "wait_any_frac": round(g("SQ_WAIT_ANY") / wc, 3), "wait_inst_any_frac": round(g("SQ_WAIT_INST_ANY") / wc, 3),
                "active_inst_any_frac": round(g("SQ_ACTIVE_INST_ANY") / wc, 3),
            }
    res["workloads"][wname] = per
print(json.dumps(res, indent=1))
