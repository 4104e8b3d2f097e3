"""Compiler's resource usage (VGPRs, SGPRs, scratch, VGPR and SGPR spills, LDS, occupancy) of every shipped kernel instantiation
(hipcc -Rpass-analysis=kernel-resource-usage over the .hip files) -> profiles/r4_kernel_resources.txt — and, for every k_paths
instantiation, WHERE its spill code sits: the assembly (-S) is searched for scratch_load/scratch_store (VGPR
spills) and v_readlane/v_writelane (SGPR spills parked in VGPR lanes); the step group is the kernel's largest basic block (the
unrolled RT_PATHS_STEPS_PER_CHECK repetitions of the four ray slots).  The script exits non-zero if a NON-counting instantiation
(the ones that ship in a timed frame) has such an instruction inside its step group.

    python tools/kernel_resources.py       # in the container: hipcc cross-compiles, no GPU needed
"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
files = ["rt_paths.hip", "rt_frame.hip", "rt_persist.hip", "rt_kernels.hip", "rt_post.hip"]
FLAGS = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-I", os.path.join(ROOT, "include"),
         "--cuda-device-only"]


def demangle(name):
    return subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip().split("(")[0].replace("void ", "")


rows = []
for f in files:
    r = subprocess.run(FLAGS + ["-c", os.path.join(ROOT, "raytrace_amd", "csrc", f), "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"],
                       capture_output=True, text=True)
    cur = None
    for l in r.stderr.splitlines():
        if "remark:" not in l:
            continue
        body = l.split("remark:", 1)[1].replace("[-Rpass-analysis=kernel-resource-usage]", "").strip()
        if body.startswith("Function Name:"):
            name = body.split(":", 1)[1].strip()
            cur = {"name": demangle(name), "mangled": name, "file": f}
            rows.append(cur)
        elif cur is not None and ":" in body:
            k, v = body.split(":", 1)
            cur[k.strip()] = v.strip()

# ---- where the spill code of the k_paths instantiations sits
SPILL = re.compile(r"^\s*(scratch_load|scratch_store|v_readlane_b32|v_writelane_b32)")
where = {}
with tempfile.TemporaryDirectory() as tmp:
    asm = os.path.join(tmp, "rt_paths.s")
    subprocess.run(FLAGS + ["-S", os.path.join(ROOT, "raytrace_amd", "csrc", "rt_paths.hip"), "-o", asm], check=True, capture_output=True)
    lines = open(asm).read().split("\n")
starts = [i for i, l in enumerate(lines) if re.match(r"^_ZN3rtd7k_paths\w+:", l)]
for s in starts:
    mangled = lines[s].split(":")[0]
    e = next(i for i in range(s, len(lines)) if "s_endpgm" in lines[i])
    # basic blocks: label lines (.LBBx_y:) split the body
    blocks, cur_label, cur_lines = [], "entry", []
    for l in lines[s + 1:e + 1]:
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            blocks.append((cur_label, cur_lines)); cur_label, cur_lines = m.group(1), []
        elif l.strip() and not l.strip().startswith((";", ".")):
            cur_lines.append(l)
    blocks.append((cur_label, cur_lines))
    valu = lambda ls: sum(1 for l in ls if l.strip().startswith("v_"))
    step_label, step_lines = max(blocks, key=lambda b: valu(b[1]))
    spill_blocks = {}
    for label, ls in blocks:
        n = sum(1 for l in ls if SPILL.match(l))
        if n:
            spill_blocks[label] = n
    where[mangled] = {"step_block": step_label, "step_valu": valu(step_lines), "in_step": spill_blocks.get(step_label, 0),
                      "elsewhere": {k: v for k, v in spill_blocks.items() if k != step_label}}

out = os.path.join(ROOT, "profiles", "r4_kernel_resources.txt")
bad = []
with open(out, "w") as fh:
    fh.write("# hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -Rpass-analysis=kernel-resource-usage (tools/kernel_resources.py)\n")
    fh.write("%-14s %-58s %5s %5s %8s %6s %6s %10s %4s\n" % ("file", "kernel", "VGPR", "SGPR", "scratchB", "vspill", "sspill", "LDS bytes", "occ"))
    for c in rows:
        fh.write("%-14s %-58s %5s %5s %8s %6s %6s %10s %4s\n" % (c["file"], c["name"][:58], c.get("VGPRs"), c.get("TotalSGPRs"), c.get("ScratchSize [bytes/lane]"),
                                                              c.get("VGPRs Spill"), c.get("SGPRs Spill"), c.get("LDS Size [bytes/block]"), c.get("Occupancy [waves/SIMD]")))
    fh.write("\n# k_paths: where the spill code sits (scratch_load/store = VGPR spills, v_readlane/v_writelane = SGPR spills kept in VGPR lanes).\n"
             "# step group = the instantiation's largest basic block; `in step` must be 0 for the non-counting (<false, ...>) builds.\n")
    fh.write("%-42s %-12s %9s %8s  %s\n" % ("kernel", "step block", "step VALU", "in step", "spill instructions in other blocks (cold: pass branches, epilogue)"))
    for c in rows:
        w = where.get(c["mangled"])
        if not w:
            continue
        counting = "k_paths<true" in c["name"]      # counting builds (never timed): totals only
        listing = ("%d in %d blocks" % (sum(w["elsewhere"].values()), len(w["elsewhere"])) if counting
                   else ", ".join("%s:%d" % kv for kv in sorted(w["elsewhere"].items()))) or "-"
        fh.write("%-42s %-12s %9d %8d  %s\n" % (c["name"][:42], w["step_block"], w["step_valu"], w["in_step"], listing))
        if "k_paths<false" in c["name"] and w["in_step"]:
            bad.append(c["name"])
print(open(out).read())
if bad:
    sys.exit("spill code inside the step group of a shipped instantiation: " + ", ".join(bad))
