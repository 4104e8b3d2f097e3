import json, sys
for line in sys.stdin:
    line = line.strip()
    if line.startswith("{"):
        d = json.loads(line)
        r = d["roofline"]
        print(d["config"]["kernel"], "cache" if d["config"].get("primary_cache") else "", "ms", d["ms_per_step"], "Mrays/s", d["value"],
              "achieved GB/s", r["achieved"], "frac", r["frac"], "kernel ms", r["avg_launch_ms"], "x", r["launches_per_frame"])
