"""CPU checks of bench.py's host-side helpers that need no GPU: the sweep of the in-kernel launch timeline (SVO_TIMELINE=1, `--timeline`)."""
import numpy as np

import bench


class _FakeCtx:
    """what hip.Context.timeline() returns: (frame counter, [16 frames][32 kinds][8 aux][2] ticks of 10 ns, t0 = 2**64 - 1 where nothing ran)"""
    def __init__(self, launches):
        self.a = np.full((16, 32, 8, 2), np.uint64(0xFFFFFFFFFFFFFFFF), np.uint64)
        self.a[..., 1] = 0
        for frame, kind, aux, t0_us, t1_us in launches:
            self.a[frame % 16, bench.TIMELINE_KINDS.index(kind), aux] = (int(t0_us * 100), int(t1_us * 100))

    def timeline(self, reset=False):
        return 1, self.a


def test_timeline_sweep_counts_queues_gaps_and_lonely_kernels():
    """Two contexts.  Detect stream (queue 0): ctx 0 runs fast 0-100 us and select 110-120 us (a 10 us gap), ctx 1 runs fast 120-220 us.
    Stage 3-5 streams: ctx 0's describe 100-200 us (queue 1), ctx 1's gauss_newton 150-250 us (queue 2).
    Busy 0-250 us:  [0,100) fast alone | [100,110) describe alone | [110,120) select + describe | [120,150) fast + describe |
    [150,200) fast + describe + gn | [200,220) fast + gn | [220,250) gn alone."""
    c0 = _FakeCtx([(1, "fast", 0, 0, 100), (1, "select", 0, 110, 120), (1, "describe", 1, 100, 200)])
    c1 = _FakeCtx([(1, "fast", 0, 120, 220), (1, "gauss_newton", 0, 150, 250)])
    r = bench.timeline_report([c0, c1], True)
    assert r["launches"] == 5 and abs(r["span_ms"] - 0.25) < 1e-9 and r["idle_frac"] == 0
    q = r["queues_executing_frac_of_busy"]
    assert abs(q["1"] - 140 / 250) < 1e-3 and abs(q["2"] - 60 / 250) < 1e-3 and abs(q["3"] - 50 / 250) < 1e-3
    assert abs(r["mean_queues_executing"] - (140 + 120 + 150) / 250) < 2e-3
    alone = r["alone_on_chip_by_kernel_frac_of_busy"]
    assert abs(alone["fast"] - 0.4) < 1e-3 and abs(alone["describe"] - 0.04) < 1e-3 and abs(alone["gauss_newton"] - 0.12) < 1e-3
    pairs = r["two_in_flight_pairs_frac_of_busy"]
    assert abs(pairs["describe + fast"] - 30 / 250) < 1e-3 and abs(pairs["describe + select"] - 10 / 250) < 1e-3 and abs(pairs["fast + gauss_newton"] - 20 / 250) < 1e-3
    det = r["per_queue"]["detect"]
    assert det["launches"] == 3 and abs(det["executing_frac_of_span"] - 210 / 250) < 1e-3 and abs(det["gap_us_mean"] - 10.0) < 1e-6      # one gap of 10 us (back-to-back launches are not gaps)
    assert r["per_queue"]["rest0"]["launches"] == 1 and r["per_queue"]["rest1"]["launches"] == 1
    assert r["hull_us_mean"]["describe"] == 100.0 and r["hull_us_mean"]["fast[0]"] == 100.0
    # with the NMS / description kept on the detect stream (post_on_rest = False) describe belongs to queue 0
    r0 = bench.timeline_report([c0, c1], False)
    assert r0["per_queue"]["detect"]["launches"] == 4 and r0["per_queue"]["rest0"]["launches"] == 0


def test_timeline_sweep_without_records_says_so():
    assert "error" in bench.timeline_report([_FakeCtx([])], True)


def test_frames_needed_and_lane_frame_never_repeat_a_frame():
    T, B, steps = 8, 192, 105
    F = bench.frames_needed(B, T, steps)
    for lane in (0, 7, 8, 100, 191):
        seen = [bench.lane_frame(lane, i, T, F) for i in range(steps)]
        assert len(set(seen)) == steps and all(j == lane % T for j, _ in seen)
        assert all(b[1] == a[1] + 1 for a, b in zip(seen, seen[1:]))            # consecutive frames of one trajectory
