#!/usr/bin/env python3
"""Write a synthetic stereo sequence file for tools/demo_stereo_odometry (format in its header comment)."""
import os, struct, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stereo_vo_amd.synth import SyntheticStereoWorld


def write_sequence(path, w=640, h=480, f=400.0, baseline=0.12, seed=11, n_frames=20, device="cpu"):
    world = SyntheticStereoWorld(w, h, f, baseline, seed=seed, n_frames=n_frames, device=device)
    with open(path, "wb") as fo:
        fo.write(b"SVOSEQ1\0" + struct.pack("<iiidddd", w, h, n_frames, f, world.cx, world.cy, baseline))
        for t in range(n_frames):
            L, R = world.render(t)
            fo.write(L.cpu().numpy().tobytes()); fo.write(R.cpu().numpy().tobytes())
    return world


if __name__ == "__main__":
    write_sequence(sys.argv[1], *(int(a) for a in sys.argv[2:4])) if len(sys.argv) > 3 else write_sequence(sys.argv[1])
