"""The replay helper behind bench.py's parity_probe / cpu_baseline legs (oracle/probe.py): independent oracle instances
on several host threads give exactly what one instance gives alone."""
import os
import numpy as np

from stereo_vo_amd.abi import StereoCamera, north_star_params


def test_threaded_replay_equals_single_replay(golden_dir):
    from oracle import oracle as O, probe as PR
    g = np.load(os.path.join(golden_dir, "oracle_small_seq.npz"))
    cam = StereoCamera.simple(float(g["F"]), float(g["cx"]), float(g["cy"]), float(g["baseline"]), int(g["W"]), int(g["H"]))
    p = north_star_params(O.default_params(), orb_nfeats=int(g["orb_nfeats"]))
    fr = [(g["L%d" % t], g["R%d" % t]) for t in range(4)]
    rev = [(b, a) for a, b in fr]
    order = [0, 1, 2, 3, 2, 1]
    a, dt = PR.replay(p, cam, fr, order)
    many, _ = PR.replay_many(p, cam, {0: fr, 1: rev, 2: fr}, order, threads=3)
    assert dt > 0 and set(many) == {0, 1, 2}
    assert all(x == y and np.array_equal(x.pose, y.pose) for x, y in zip(a, many[0]))
    assert all(x == y for x, y in zip(many[0], many[2]))
    assert any(not (x == y) for x, y in zip(many[0], many[1]))
    # against the committed vectors: the digests are digests of the same lists
    assert a[1].n[0] == len(g["kps0_1"]) and a[3].n[2] == len(g["matches3"])
    lists, flags, et, er = PR.compare(a[2], many[2][2])
    assert lists and flags and et == 0.0 and er == 0.0
