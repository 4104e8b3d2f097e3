"""When do the WORKGROUPS of a k_paths launch end?  A CU is free for the next launch only when the last of its 16 waves has ended, so
what two launches on two streams can overlap is (kernel end - workgroup end), not the waves' own drain.  Needs the diagnostic
build:  tools/variant.sh wt rt_paths.hip -DRT_DIAG_WAVE_TIMES ;  RT_AMD_LIB=$PWD/raytrace_amd/librt_amd_wt.so python tools/lab/r4/wg_end_times.py [r/N] [W H spp depth]"""
import os, subprocess, sys
import numpy as np
dump = "/tmp/rt_wave_dump.bin"
env = dict(os.environ, RT_DEBUG_WAVE_DUMP=dump)
here = os.path.dirname(os.path.abspath(__file__))
out = subprocess.run([sys.executable, os.path.join(here, "..", "..", "drain_times.py")] + sys.argv[1:], env=env, capture_output=True, text=True)
print(out.stdout.strip())
rec = np.fromfile(dump, dtype=np.uint64).reshape(-1, 4).astype(np.int64)
rec = rec[rec[:, 2] != 0]
t0 = rec[:, 0].min()
start, exh, end, wg = (rec[:, 0] - t0) / 100.0, (rec[:, 1] - t0) / 100.0, (rec[:, 2] - t0) / 100.0, rec[:, 3]
kend = end.max()
nwg = int(wg.max()) + 1
wg_end = np.array([end[wg == g].max() for g in range(nwg)])
wg_first = np.array([end[wg == g].min() for g in range(nwg)])
wg_start = np.array([start[wg == g].min() for g in range(nwg)])
q = lambda a: "min %.0f p10 %.0f median %.0f p90 %.0f max %.0f" % (a.min(), np.percentile(a, 10), np.median(a), np.percentile(a, 90), a.max())
print("waves %d, workgroups %d, kernel span %.0f us (100 MHz clock)" % (len(rec), nwg, kend))
print("wave out of paths at: " + q(exh[exh > 0]))
print("wave end:             " + q(end))
print("workgroup start:      " + q(wg_start))
print("workgroup end:        " + q(wg_end))
print("CU idle between its workgroup's end and the kernel's end: mean %.1f us (%.2f %% of the span) — what a second stream's launch can take" %
      ((kend - wg_end).mean(), 100.0 * (kend - wg_end).mean() / kend))
print("wave-slots idle between a wave's end and its workgroup's end: mean %.1f us — NOT reusable (the workgroup holds the CU's LDS)" %
      np.mean([wg_end[g] - end[wg == g].mean() for g in range(nwg)]))
print("waves' own drain (out of paths -> end): mean %.1f us" % (end - exh)[exh > 0].mean())
