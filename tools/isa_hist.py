"""Instruction mix of k_paths' step group and what it costs to issue: compiles rt_paths.hip to gfx950 assembly
(hipcc -S), takes the shipped headline instantiation k_paths<false, 0, 8, true>, finds the step group (its largest basic block:
RT_PATHS_STEPS_PER_CHECK repetitions of the four ray slots, ten slot-steps with the default 3 / 0x3), counts the opcodes and
prices every VALU opcode with the issue cost measured by tools/ubench/valu_rate (profiles/r3_ubench_valu_rate.txt, the
4-waves-per-SIMD column — the occupancy the kernel runs at).  Writes profiles/r4_step_loop_isa_hist.json; bench.py reads
`cycles_per_valu_inst` from it for roofline.valu.pipe_busy_weighted.

    python tools/isa_hist.py            # run in the container (hipcc cross-compiles, no GPU needed)
"""
import collections, hashlib, json, os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNEL = "_ZN3rtd7k_pathsILb0ELi0ELi8ELb1EEE"   # k_paths<false, 0, 8, true>
UBENCH = os.path.join(ROOT, "profiles", "r3_ubench_valu_rate.txt")
# opcodes the micro-benchmark did not time, priced like the measured opcode of the same class (stated in the output)
ALIAS = {"v_cmp": "v_cmp_eq_u32 e64", "v_cndmask_b32": "v_cndmask e64(sgpr)", "v_sub_f32": "v_add_f32", "v_lshl_add_u32": "v_lshl_add_u32",
         "v_min_f32": "v_max_f32", "v_min_u32": "v_min_u32", "v_min3_u32": "v_min3_f32", "v_mov_b64": "v_pk_add_f32", "v_cvt_f32_i32": "v_cvt_f32_u32",
         "v_bfe_i32": "v_bfe_u32", "v_ashrrev_i32": "v_lshrrev_b32", "v_subrev_u32": "v_sub_u32", "v_mul_lo_u32": "v_mul_lo_u32"}


def source_sha16():
    h = hashlib.sha256()
    d = os.path.join(ROOT, "raytrace_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".hpp")) and name != "rt_api.hip":
            h.update(open(os.path.join(d, name), "rb").read())
    h.update(open(os.path.join(ROOT, "include", "rt_math.h"), "rb").read())
    return h.hexdigest()[:16]


def issue_costs():
    cost = {}
    for line in open(UBENCH):
        m = re.match(r"^(\S.*?)\s+1w:.*4w:\s*[\d.]+ /wave\s+([\d.]+) /SIMD", line)
        if m:
            cost[m.group(1).strip()] = float(m.group(2))
    return cost


def cost_of(op, cost):
    base = re.sub(r"_(e32|e64|sdwa|dpp)$", "", op)
    for key in (base, base.replace("_b32", "").replace("_e64", "")):
        if key in cost:
            return cost[key], key
    if base.startswith("v_cmp"):
        return cost[ALIAS["v_cmp"]], ALIAS["v_cmp"]
    if base in ALIAS and ALIAS[base] in cost:
        return cost[ALIAS[base]], ALIAS[base]
    return None, None


def main():
    with tempfile.TemporaryDirectory() as tmp:
        asm = os.path.join(tmp, "rt_paths.s")
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
               "-I", os.path.join(ROOT, "include"), "-S", "--cuda-device-only", os.path.join(ROOT, "raytrace_amd", "csrc", "rt_paths.hip"), "-o", asm]
        subprocess.run(cmd, check=True, capture_output=True)
        lines = open(asm).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith(KERNEL) and l.rstrip().endswith(":") or l.startswith(KERNEL) and ": ;" in l)
    end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
    blocks, cur = [], None
    for l in lines[start:end + 1]:
        if re.match(r"^(\.LBB\d+_\d+):", l) or l.startswith("; %bb."):
            cur = {"label": l.split(":")[0].strip(), "ops": []}
            blocks.append(cur)
        elif cur is not None:
            m = re.match(r"^\s+([a-z][a-z0-9_]+)\b", l)
            if m:
                cur["ops"].append(m.group(1))
    step = max(blocks, key=lambda b: sum(o.startswith("v_") for o in b["ops"]))
    cost = issue_costs()
    hist = collections.Counter(step["ops"])
    valu = {o: n for o, n in hist.items() if o.startswith("v_")}
    table, cycles, unpriced = [], 0.0, []
    for o, n in sorted(valu.items(), key=lambda kv: -kv[1]):
        c, src = cost_of(o, cost)
        if c is None:
            unpriced.append(o); c, src = 3.4, "unmeasured: priced at 3.4"
        table.append({"opcode": o, "count": n, "issue_cycles_each": c, "priced_as": src})
        cycles += n * c
    nvalu = sum(valu.values())
    whole = collections.Counter(o for b in blocks for o in b["ops"])
    steps_per_check = 3
    shadow_reps = 0x3
    slot_steps = sum(4 if (shadow_reps >> r) & 1 else 2 for r in range(steps_per_check))
    out = {
        "kernel": "k_paths<false, 0, 8, true>", "kernel_source_sha16": source_sha16(),
        "tool": "tools/isa_hist.py (hipcc -S; issue costs: profiles/r3_ubench_valu_rate.txt, 4 waves per SIMD)",
        "step_group_block": step["label"], "slot_steps_per_group": slot_steps,
        "valu_insts": nvalu, "valu_insts_per_slot_step": round(nvalu / slot_steps, 2),
        "salu_insts": sum(n for o, n in hist.items() if o.startswith("s_") and o not in ("s_waitcnt", "s_nop")),
        "lds_insts": sum(n for o, n in hist.items() if o.startswith("ds_")), "vmem_insts": sum(n for o, n in hist.items() if o.startswith(("buffer_", "global_", "flat_"))),
        "valu_issue_cycles": round(cycles, 1), "cycles_per_valu_inst": round(cycles / nvalu, 4), "issue_cycles_per_slot_step": round(cycles / slot_steps, 1),
        "select_insts": sum(n for o, n in valu.items() if o.startswith("v_cndmask")), "compare_insts": sum(n for o, n in valu.items() if o.startswith("v_cmp")),
        "unpriced_opcodes": unpriced, "valu_opcodes": table,
        "whole_kernel": {"valu_insts": sum(n for o, n in whole.items() if o.startswith("v_")), "basic_blocks": len(blocks)},
    }
    path = os.path.join(ROOT, "profiles", "r4_step_loop_isa_hist.json")
    json.dump(out, open(path, "w"), indent=1)
    print(json.dumps({k: v for k, v in out.items() if k != "valu_opcodes"}, indent=1))
    print("%-24s %5s %7s" % ("opcode", "count", "cycles"))
    for r in table:
        print("%-24s %5d %7.2f  (%s)" % (r["opcode"], r["count"], r["issue_cycles_each"], r["priced_as"]))


if __name__ == "__main__":
    main()
