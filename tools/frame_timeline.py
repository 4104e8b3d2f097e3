"""Where a small frame's time goes between its kernels: rocprofv3 --kernel-trace of a bench run -> per frame, each kernel's duration
and the gap before it.   python tools/frame_timeline.py <dir with *_kernel_trace.csv> [frames_to_skip]"""
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
short = lambda n: n.split("(")[0].replace("void ", "").replace("rtd::", "")[:28]
# the timed frames: the last 40 occurrences of the frame's first kernel (a k_primary*)
starts = [i for i, r in enumerate(rows) if "k_primary" in r["Kernel_Name"]]
starts = starts[-41:]
dur = collections.defaultdict(list); gap = collections.defaultdict(list); total = []
for a, b in zip(starts[:-1], starts[1:]):
    fr = rows[a:b]
    for i, r in enumerate(fr):
        k = "%d %s" % (i, short(r["Kernel_Name"]))
        dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
        prev_end = int(rows[a + i - 1]["End_Timestamp"]) if a + i > 0 else int(r["Start_Timestamp"])
        gap[k].append((int(r["Start_Timestamp"]) - prev_end) / 1e3)
    total.append((int(rows[b]["Start_Timestamp"]) - int(rows[a]["Start_Timestamp"])) / 1e3)
print("frames %d, frame period %.1f us (median)" % (len(total), sorted(total)[len(total) // 2]))
for k in dur:
    d, g = sorted(dur[k]), sorted(gap[k])
    print("  %-32s runs %5.1f us   idle before it %5.1f us   (in %d frames)" % (k, d[len(d) // 2], g[len(g) // 2], len(d)))
