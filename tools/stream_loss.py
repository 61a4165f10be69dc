#!/usr/bin/env python3
"""Why do some streams of bench.py's long leg end up invalid?  

Runs the long leg's schedule (192 streams = 8 trajectories x 24 phases through the 'street' world, warm-up + 100 steps) on the HIP path with a
synchronisation and a read-back of every stream's result record after EVERY step -- the records carry the stage-4 funnel
(svo_result.track_stats, bit-exact with the oracle's: tests/test_gpu_parity.py) -- and prints, for every stream that is invalid at the last
step: the frame it failed at, the error code, the funnel around that frame, the ground-truth motion of that frame, and what the OTHER
streams that played the same frame of the same trajectory (another phase, i.e. another previous frame? no: the same pair of frames, later
in their own run) did with it.  Product path + renderer only; nothing here reads oracle/.

    python tools/stream_loss.py [--steps 105] [--out gpurun_out/stream_loss.json]
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from stereo_vo_amd import hip  # noqa: E402
from stereo_vo_amd.abi import TS_NAMES, north_star_params  # noqa: E402
from stereo_vo_amd.pipeline import StreamBatch  # noqa: E402
from stereo_vo_amd.synth import SyntheticStereoWorld  # noqa: E402

EC = {0: "none", 1: "bad_cond_number", 2: "incr_cost_stg1", 3: "incr_cost_stg2", 4: "first_iteration", 5: "bad_tracking"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=105)
    ap.add_argument("--lanes", type=int, default=192)
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    W, H, B, NC, T = 1280, 960, a.lanes, 3, 8
    dev = torch.device("cuda", 0)
    F = bench.frames_needed(B, T, a.steps)
    worlds = [SyntheticStereoWorld(W, H, 800.0, 0.12, seed=j, n_frames=F, device=dev, scene_seed=j % 4, scene="street", noise_on_device=True) for j in range(T)]
    frames = [[w.render(t) for t in range(F)] for w in worlds]
    cam = worlds[0].camera()
    p = north_star_params(hip.default_params(), orb_nfeats=2000)
    batch = StreamBatch(p, cam, W, H, B, NC, device=0)
    hist = []                                                   # [step][lane] -> (valid, error, kps, matches, track_stats[8], frame)
    for i in range(a.steps):
        ptrs, fr = [], []
        for l in range(B):
            j, t = bench.lane_frame(l, i, T, F)
            L_, R_ = frames[j][t]
            ptrs.append((L_.data_ptr(), R_.data_ptr())); fr.append((j, t))
        batch.step(ptrs)
        res = batch.results()
        hist.append([(int(r.valid), int(r.error_code), int(r.detected_left[0]), int(r.stereo_matches[0]), [int(v) for v in r.track_stats], fr[l]) for l, r in enumerate(res)])
    last = hist[-1]
    dead = [l for l in range(B) if not last[l][0]]
    # what every (trajectory, frame) pair did on every stream that played it while healthy: index by (j, t)
    by_frame = {}
    for i in range(1, a.steps):
        for l in range(B):
            v, ec, k, m, ts, (j, t) = hist[i][l]
            pv = hist[i - 1][l]
            if pv[0] or i == 1:                                 # the stream came into this frame healthy
                by_frame.setdefault((j, t), []).append((l, v, ec, ts))
    report = {"streams": B, "steps": a.steps, "invalid_at_last_step": len(dead), "lanes": []}
    for l in dead:
        # first step from which the stream never is valid again
        i_fail = a.steps - 1
        while i_fail > 1 and not hist[i_fail - 1][l][0]:
            i_fail -= 1
        v, ec, k, m, ts, (j, t) = hist[i_fail][l]
        gt = worlds[j].gt_delta(t) if t > 0 else np.eye(4)
        step_m = float(np.linalg.norm(gt[:3, 3])); rot = float(np.arccos(np.clip((np.trace(gt[:3, :3]) - 1) / 2, -1, 1)))
        others = [(o[0], o[1], EC.get(o[2], o[2]), o[3][0], o[3][6], o[3][7]) for o in by_frame.get((j, t), []) if o[0] != l]
        window = []
        for i in range(max(1, i_fail - 3), min(a.steps, i_fail + 3)):
            vv, ee, kk, mm, tt, ff = hist[i][l]
            window.append({"step": i, "frame": ff[1], "valid": vv, "error": EC.get(ee, ee), "kps": kk, "matches": mm, **dict(zip(TS_NAMES, tt))})
        report["lanes"].append({"lane": l, "trajectory": j, "failed_at_step": i_fail, "frame": t, "error": EC.get(ec, ec), "gt_step_m": round(step_m, 4), "gt_rot_rad": round(rot, 5),
                                "funnel_around_failure": window,
                                "same_frame_on_other_streams": {"n": len(others), "valid": sum(o[1] for o in others), "tracked_median": float(np.median([o[5] for o in others])) if others else None,
                                                                "candidates_median": float(np.median([o[3] for o in others])) if others else None}})
    # the healthy population, for scale
    ok = [hist[i][l] for i in range(2, a.steps) for l in range(B) if hist[i][l][0]]
    report["healthy_frames"] = {"n": len(ok), **{nm: round(float(np.mean([o[4][q] for o in ok])), 1) for q, nm in enumerate(TS_NAMES)},
                                "tracked_p05": float(np.percentile([o[4][7] for o in ok], 5)), "tracked_min": int(min(o[4][7] for o in ok)),
                                "gt_step_m_mean": round(float(np.mean([np.linalg.norm(worlds[o[5][0]].gt_delta(o[5][1])[:3, 3]) for o in ok[::97]])), 4)}
    fails_by_frame = {}
    for e in report["lanes"]:
        fails_by_frame.setdefault("traj %d frame %d" % (e["trajectory"], e["frame"]), []).append(e["lane"])
    report["failures_by_trajectory_frame"] = fails_by_frame
    batch.close()
    txt = json.dumps(report, indent=1)
    if a.out:
        open(a.out, "w").write(txt)
    print("invalid at last step: %d of %d" % (len(dead), B))
    print("failures by (trajectory, frame):", json.dumps(fails_by_frame))
    print("healthy frames:", json.dumps(report["healthy_frames"]))
    for e in report["lanes"]:
        print("lane %3d traj %d failed at step %3d frame %3d: %s; gt step %.3f m %.4f rad; same frame on %d other streams: %d valid, tracked median %s" %
              (e["lane"], e["trajectory"], e["failed_at_step"], e["frame"], e["error"], e["gt_step_m"], e["gt_rot_rad"], e["same_frame_on_other_streams"]["n"],
               e["same_frame_on_other_streams"]["valid"], e["same_frame_on_other_streams"]["tracked_median"]))
        for wrow in e["funnel_around_failure"]:
            print("      ", json.dumps(wrow))


if __name__ == "__main__":
    main()
