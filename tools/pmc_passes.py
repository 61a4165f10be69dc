#!/usr/bin/env python3
"""Run ON THE GPU BOX (through gpurun): hardware counters of every kernel of this repo, per launch, steady state.

Separate `rocprofv3 --pmc` passes of one bench.py command (kernel-trace only; never combined with sys / runtime traces):
  rd   TCC_EA0_RDREQ_32B / _64B / _128B   -> read bytes  = 32 n32 + 64 n64 + 128 n128
  wr   TCC_EA0_WRREQ / _64B               -> write bytes = 64 n64 + 32 (n - n64)
  sq1  SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES SQ_WAVE_CYCLES
  sq2  SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAVES SQ_INSTS_MFMA GRBM_GUI_ACTIVE
Counting the L2's memory-side requests BY SIZE is the gfx950 correction /opt/skills/guides/MI355X_MICROARCH.md asks for
(FETCH_SIZE tallies every read request at 64 B; on this path nearly every request is a 128-byte line).

The launches of the first `--skip-steps` steps are dropped (un-speculated FAST threshold, first touch).  Derived per kernel:
  valu_issue_frac = SQ_INSTS_VALU * 4 / (1024 SIMDs * cycles),  cycles = GRBM_GUI_ACTIVE / 8 (the counter sums the 8 XCDs)
  lds_conflict_frac = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE
Writes gpurun_out/<tag>_pmc.json.   Usage: python tools/pmc_passes.py <tag> [bench.py arguments ...]
"""
import collections, csv, glob, json, os, subprocess, sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _kname import kname

root = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
tag = sys.argv[1]
extra = sys.argv[2:]
STEPS, WARM, SKIP = 5, 3, 3          # 8 steps in all, the first 3 dropped
LANES = 64
cmd = [sys.executable, os.path.join(root, "bench.py"), "--steps", str(STEPS), "--warmup", str(WARM), "--cpu-frames", "0", "--long-steps", "0", "--host-fed-steps", "0",
       "--single-stream", "0", "--exclusive", "0", "--relief-lanes", "0", "--cut-steps", "0", "--other-workloads", "0", "--contexts", "1", "--lanes", str(LANES), "--frames", "60"] + extra
env = dict(os.environ, TMPDIR="/tmp")
PASSES = {
    "rd": ["TCC_EA0_RDREQ_32B_sum", "TCC_EA0_RDREQ_64B_sum", "TCC_EA0_RDREQ_128B_sum"],
    "wr": ["TCC_EA0_WRREQ_sum", "TCC_EA0_WRREQ_64B_sum"],
    "sq1": ["SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES"],
    "sq2": ["SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_LDS_IDX_ACTIVE", "SQ_INSTS_VMEM_RD", "SQ_INSTS_SMEM", "SQ_WAVES", "SQ_INSTS_MFMA", "GRBM_GUI_ACTIVE"],
    "mfma": ["SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CU_CYCLES", "SQ_INSTS_VALU_MFMA_F64", "SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_ADD_F64"],      # round 5: how busy the matrix pipe really is
}
if os.environ.get("PMC_ONLY"):          # e.g. PMC_ONLY=sq2,mfma: a subset of the passes (the summary then lacks what the others measure)
    PASSES = {k: v for k, v in PASSES.items() if k in os.environ["PMC_ONLY"].split(",")}


def run_pass(name, counters):
    try:
        return _run_pass(name, counters)
    except subprocess.TimeoutExpired:
        return {}, {}


def _run_pass(name, counters):
    out = "/tmp/pmc_%s_%s" % (tag, name)
    subprocess.run(["rm", "-rf", out])
    subprocess.run(["rocprofv3", "--pmc"] + counters + ["--kernel-trace", "--output-format", "csv", "-d", out, "--"] + cmd,
                   cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=False, timeout=400)
    files = glob.glob(out + "/**/*counter_collection.csv", recursive=True)
    if not files:
        return {}, {}
    per = collections.defaultdict(lambda: collections.defaultdict(dict))       # kernel -> dispatch id -> counter -> value
    for r in csv.DictReader(open(files[0])):
        n = kname(r["Kernel_Name"])
        if n is not None:
            d = per[n][int(r["Dispatch_Id"])]
            d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    avg, launches = {}, {}
    for n, disp in per.items():
        ids = sorted(disp)
        keep = ids[(len(ids) * SKIP) // (STEPS + WARM):]
        launches[n] = len(keep) / float(STEPS + WARM - SKIP)                      # launches per step
        avg[n] = {c: sum(disp[i].get(c, 0.0) for i in keep) / len(keep) for c in counters}
    return avg, launches


res = {"_how": "tools/pmc_passes.py: four separate rocprofv3 --pmc passes of `%s`; launches of the first %d of %d steps dropped; per-launch averages" % (" ".join(cmd[1:]), SKIP, STEPS + WARM),
       "lanes": LANES, "workload": "config2", "kernels": {}}
for a in extra:
    if a in ("config3", "config5"):
        res["workload"] = a
allc, launches = collections.defaultdict(dict), {}
for name, counters in PASSES.items():
    avg, ln = run_pass(name, counters)
    launches.update(ln)
    for k, d in avg.items():
        allc[k].update(d)
path = 0.0
for k, d in sorted(allc.items()):
    e = {"launches_per_step": round(launches.get(k, 0), 3)}
    if "TCC_EA0_RDREQ_128B_sum" in d:
        e["read_bytes"] = int(32 * d["TCC_EA0_RDREQ_32B_sum"] + 64 * d["TCC_EA0_RDREQ_64B_sum"] + 128 * d["TCC_EA0_RDREQ_128B_sum"])
    if "TCC_EA0_WRREQ_sum" in d:
        e["write_bytes"] = int(64 * d["TCC_EA0_WRREQ_64B_sum"] + 32 * (d["TCC_EA0_WRREQ_sum"] - d["TCC_EA0_WRREQ_64B_sum"]))
    path += (e.get("read_bytes", 0) + e.get("write_bytes", 0)) * launches.get(k, 0)
    cyc = d.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
    if cyc > 0 and "SQ_INSTS_VALU" in d:
        e["cycles"] = int(cyc)
        e["valu_issue_frac"] = round(d["SQ_INSTS_VALU"] * 4.0 / (1024.0 * cyc), 4)
    if cyc > 0 and "SQ_VALU_MFMA_BUSY_CYCLES" in d:
        e["mfma_busy_frac"] = round(d["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * cyc), 4)      # rocprofv3's MfmaUtil: busy cycles summed over the SIMDs / (cycles x 1024 SIMDs)
    if d.get("SQ_LDS_IDX_ACTIVE", 0) > 0 and "SQ_LDS_BANK_CONFLICT" in d:
        e["lds_conflict_frac"] = round(d["SQ_LDS_BANK_CONFLICT"] / d["SQ_LDS_IDX_ACTIVE"], 4)
    if d.get("SQ_WAVES", 0) > 0 and "SQ_INSTS_VALU" in d:
        e["valu_per_wave"] = round(d["SQ_INSTS_VALU"] / d["SQ_WAVES"], 1)
    e["counters"] = {c: float("%.5g" % v) for c, v in sorted(d.items())}
    res["kernels"][k] = e
res["path_bytes_per_step"] = int(path)
res["path_MB_per_pair"] = round(path / LANES / 1e6, 2)
os.makedirs(os.path.join(root, "gpurun_out"), exist_ok=True)
json.dump(res, open(os.path.join(root, "gpurun_out", "%s_pmc.json" % tag), "w"), indent=1)
print(json.dumps({k: {x: v[x] for x in v if x != "counters"} for k, v in res["kernels"].items()}, indent=0))
print("path_MB_per_pair", res["path_MB_per_pair"])
