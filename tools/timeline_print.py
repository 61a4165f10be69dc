"""print the `timeline` block of a bench line (bench.py --timeline N): python tools/timeline_print.py <bench.json>"""
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
t = d.get("timeline") or {}
print(sys.argv[1].split("/")[-1], d["value"], "pairs/s", d["ms_per_step"], "ms/step; GPU_MAX_HW_QUEUES", d.get("env", {}).get("GPU_MAX_HW_QUEUES"))
for k in ("steps", "launches", "span_ms", "idle_frac", "queues_executing_frac_of_busy", "mean_queues_executing", "alone_on_chip_by_kernel_frac_of_busy", "two_in_flight_pairs_frac_of_busy", "per_queue", "hull_us_mean", "error"):
    if k in t:
        print("  ", k, json.dumps(t[k]))
