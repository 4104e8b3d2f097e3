"""Where a step of k_paths spends its time ONCE THE PATHS HAVE RUN OUT (the drain): a -DRT_DIAG_STEP_TIMES build brackets the
three phases of every step of an exhausted wave with s_waitcnt + s_memtime — nibble reads from LDS / byte loads / arithmetic and
swizzle-table reads — and sums the ticks.   tools/variant.sh st rt_paths.hip -DRT_DIAG_STEP_TIMES ;
RT_AMD_LIB=$PWD/raytrace_amd/librt_amd_st.so python tools/lab/r4/step_times.py [r/N] [W H spp depth]"""
import os, re, subprocess, sys
env = dict(os.environ, _DRAIN_CHILD="1", RT_DEBUG_STATS="1")
here = os.path.dirname(os.path.abspath(__file__))
out = subprocess.run([sys.executable, os.path.join(here, "..", "..", "drain_times.py")] + sys.argv[1:], env=env, capture_output=True, text=True)
m = re.search(r"raw: loop_iters (\d+) s_lanes (\d+) f_lanes (\d+) passes (\d+) pass_lanes (\d+) s_execs (\d+) f_execs (\d+)", out.stderr)
m2 = re.search(r"raw2: sky_lanes (\d+)", out.stderr)
if not m or not m2:
    sys.exit("no counters: " + out.stderr[-2000:])
nbytes, t_mem, _, _, _, t_lds, steps = (int(x) for x in m.groups())
t_alu = int(m2.group(1))
tot = t_lds + t_mem + t_alu
print("timed steps of exhausted waves: %d (%.1f per wave) | ticks per step: LDS nibble reads %.0f, byte loads %.0f, arithmetic + table reads %.0f, total %.0f "
      "| lanes fetching a byte per step %.2f" % (steps, steps / 4096.0, t_lds / steps, t_mem / steps, t_alu / steps, tot / steps, nbytes / steps))
