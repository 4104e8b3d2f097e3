"""Compiler's resource usage (VGPRs, SGPRs, spills, LDS, occupancy) of every shipped kernel instantiation:
hipcc -Rpass-analysis=kernel-resource-usage over the four .hip files -> profiles/r3_kernel_resources.txt."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
files = ["rt_paths.hip", "rt_persist.hip", "rt_kernels.hip", "rt_post.hip"]
rows = []
for f in files:
    r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
                        "-I", os.path.join(ROOT, "include"), "--cuda-device-only", "-c", os.path.join(ROOT, "raytrace_amd", "csrc", f),
                        "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True)
    cur = None
    for l in r.stderr.splitlines():
        if "remark:" not in l:
            continue
        body = l.split("remark:", 1)[1].replace("[-Rpass-analysis=kernel-resource-usage]", "").strip()
        if body.startswith("Function Name:"):
            name = body.split(":", 1)[1].strip()
            dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
            cur = {"name": dem.split("(")[0].replace("void ", ""), "file": f}
            rows.append(cur)
        elif cur is not None and ":" in body:
            k, v = body.split(":", 1)
            cur[k.strip()] = v.strip()
out = os.path.join(ROOT, "profiles", "r3_kernel_resources.txt")
with open(out, "w") as fh:
    fh.write("# hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -Rpass-analysis=kernel-resource-usage (tools/kernel_resources.py)\n")
    fh.write("%-14s %-58s %5s %5s %8s %7s %10s %4s\n" % ("file", "kernel", "VGPR", "SGPR", "scratchB", "spills", "LDS bytes", "occ"))
    for c in rows:
        fh.write("%-14s %-58s %5s %5s %8s %7s %10s %4s\n" % (c["file"], c["name"][:58], c.get("VGPRs"), c.get("TotalSGPRs"), c.get("ScratchSize [bytes/lane]"),
                                                          c.get("VGPRs Spill"), c.get("LDS Size [bytes/block]"), c.get("Occupancy [waves/SIMD]")))
print(open(out).read())
