#!/usr/bin/env python3
"""Run ON THE GPU BOX (through gpurun): HBM/fabric traffic of every svo kernel, per launch, from the L2's memory-side
request counters, in two separate `rocprofv3 --pmc` passes (kernel-trace only; never combined with sys/runtime traces):

  reads  = 32 * TCC_EA0_RDREQ_32B + 64 * TCC_EA0_RDREQ_64B + 128 * TCC_EA0_RDREQ_128B
  writes = 64 * TCC_EA0_WRREQ_64B + 32 * (TCC_EA0_WRREQ - TCC_EA0_WRREQ_64B)

Counting requests BY SIZE is the gfx950 correction /opt/skills/guides/MI355X_MICROARCH.md asks for: FETCH_SIZE tallies
every read request at 64 B, and on this path practically every request is a 128-byte line (so FETCH_SIZE reads 1/2).
Writes JSON to gpurun_out/pmc_traffic.json: bytes per launch, averaged over the launches of the profiled run
(k_resize: over its 7 launches per frame; k_hamming: over the left-right and tracking launches).
Usage: python tools/pmc_traffic.py [bench.py arguments ...]
"""
import collections, csv, glob, json, os, subprocess, sys

root = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
extra = sys.argv[1:]
cmd = [sys.executable, os.path.join(root, "bench.py"), "--steps", "3", "--warmup", "1", "--cpu-frames", "0"] + extra
env = dict(os.environ, TMPDIR="/tmp")


def run_pass(tag, counters):
    out = "/tmp/pmc_" + tag
    subprocess.run(["rm", "-rf", out])
    subprocess.run(["rocprofv3", "--pmc"] + counters + ["--kernel-trace", "--output-format", "csv", "-d", out, "--"] + cmd,
                   cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=False)
    files = glob.glob(out + "/**/*counter_collection.csv", recursive=True)
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(files[0])):
        n = r["Kernel_Name"].split("(")[0]
        if n.startswith("k_"):
            acc[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in acc.items()}


rd = run_pass("rd", ["TCC_EA0_RDREQ_32B_sum", "TCC_EA0_RDREQ_64B_sum", "TCC_EA0_RDREQ_128B_sum"])
wr = run_pass("wr", ["TCC_EA0_WRREQ_sum", "TCC_EA0_WRREQ_64B_sum"])
res = {"_how": __doc__.strip().split("\n\n")[0] + " See tools/pmc_traffic.py.", "command": " ".join(cmd[1:]), "read_bytes": {}, "write_bytes": {}}
for k, d in rd.items():
    res["read_bytes"][k] = int(32 * d.get("TCC_EA0_RDREQ_32B_sum", 0) + 64 * d.get("TCC_EA0_RDREQ_64B_sum", 0) + 128 * d.get("TCC_EA0_RDREQ_128B_sum", 0))
for k, d in wr.items():
    n, n64 = d.get("TCC_EA0_WRREQ_sum", 0), d.get("TCC_EA0_WRREQ_64B_sum", 0)
    res["write_bytes"][k] = int(64 * n64 + 32 * (n - n64))
os.makedirs(os.path.join(root, "gpurun_out"), exist_ok=True)
json.dump(res, open(os.path.join(root, "gpurun_out", "pmc_traffic.json"), "w"), indent=1)
print(json.dumps(res))
