#!/usr/bin/env python3
"""One line per kernel of a tools/pmc_passes.py summary (gpurun_out/<tag>_pmc.json), heaviest VALU user first."""
import json, sys
d = json.load(open(sys.argv[1]))
tot = 0.0
for k, v in sorted(d["kernels"].items(), key=lambda kv: -kv[1]["counters"].get("SQ_INSTS_VALU", 0) * kv[1].get("launches_per_step", 1)):
    c = v["counters"]; n = v.get("launches_per_step", 1); tot += c.get("SQ_INSTS_VALU", 0) * n
    print("%-24s valu %8.2fM x%.1f  /wave %7.1f  issue %.2f  lds_conf %.2f  mfma %7.2fM  cycles %8d  rd %.1f MB wr %.1f MB"
          % (k, c.get("SQ_INSTS_VALU", 0) / 1e6, n, v.get("valu_per_wave", 0), v.get("valu_issue_frac", 0), v.get("lds_conflict_frac", 0), c.get("SQ_INSTS_MFMA", 0) / 1e6, v.get("cycles", 0), v.get("read_bytes", 0) / 1e6, v.get("write_bytes", 0) / 1e6))
print("total VALU wave-instructions per context-step: %.1f M; path %.2f MB/pair" % (tot / 1e6, d.get("path_MB_per_pair", 0)))
