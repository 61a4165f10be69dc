"""Python face of the schedulers of include/svo_batch.h (stereo_vo_amd/csrc/svo_batch.cpp).

The scheduling itself -- several batched contexts side by side on ONE GPU, the detect stream / stage 3-5 stream choreography
with its events and priorities, the frame dealer of the frame-parallel stream -- lives in libsvo_hip.so behind the C-ABI, so
that a C / C++ host gets it with one call per step (tools/batch_streams.cpp drives the benchmarked shape that way).  This
module only adapts it to torch: device buffers as tensors, torch streams / events as the raw handles the C calls take.

  StreamBatch          svo_batch: `contexts` contexts of lanes / contexts streams each; the step bench.py times
  FrameParallelStream  svo_fpstream: one stream's consecutive frames dealt round-robin to `contexts` contexts

Product path only: nothing here touches oracle/.  The reference has no counterpart (one estimator, one thread:
libstereo-odometry.h:732-831 is all per-instance state, which is what makes the lanes independent).
"""
import ctypes as C

import torch

from . import hip
from .abi import Result


def _frames(ptrs, w, h, stride):
    fr = (hip.Frame * len(ptrs))()
    for i, (l, r) in enumerate(ptrs):
        fr[i].left = hip.Image(l, w, h, stride)
        fr[i].right = hip.Image(r, w, h, stride)
    return fr


class StreamBatch:
    def __init__(self, params, cam, width, height, lanes, contexts=1, device=0, schedule="pipelined", post_on_rest=True,
                 det_priority="high", kernel_times=False, max_octaves=1, max_kps=4096, max_cand=None, det_streams=1, rest_streams=0, detect_ahead=True):
        assert contexts >= 1 and lanes % contexts == 0 and lanes // contexts <= hip.MAX_LANES, \
            "lanes must split evenly over the contexts, at most %d streams per context" % hip.MAX_LANES
        self.L = hip.lib()
        self.W, self.H, self.B, self.NC, self.Bc = width, height, lanes, contexts, lanes // contexts
        self.dev = torch.device("cuda", device)
        self.pipelined = contexts > 1 and schedule == "pipelined"
        if max_cand is None:
            max_cand = (1 << 18) if width * height > 2000000 else (1 << 17)
        cfg = hip.BatchConfig()
        self.L.svo_batch_config_defaults(C.byref(cfg))
        c = cfg.ctx
        c.device, c.n_lanes, c.max_w, c.max_h, c.max_kps, c.max_cand = device, self.Bc, width, height, max_kps, max_cand
        c.kernel_times, c.max_octaves = int(kernel_times), int(max_octaves)
        cfg.n_contexts = contexts
        cfg.schedule = 0 if schedule == "pipelined" else 1
        cfg.det_priority_high = int(det_priority == "high")
        # post_on_rest: False = NMS + describe stay on the detect stream; True = on the stage 3-5 stream; "own" = on a third stream
        cfg.post_mode = 2 if post_on_rest == "own" else (3 if post_on_rest == "select" else int(bool(post_on_rest)))
        cfg.det_streams = max(1, det_streams)
        cfg.rest_streams = max(0, rest_streams)          # 0 = one stage 3-5 stream per context
        cfg.no_detect_ahead = int(not detect_ahead)      # False: a context's detector waits for its whole previous frame (rounds 1-3)
        h = C.c_void_p()
        rc = self.L.svo_batch_create_sized(C.byref(cfg), C.c_size_t(C.sizeof(cfg)), C.byref(h))      # a stale mirror of svo_batch_config is refused by the library
        self.h = h
        if rc != 0:
            msg = self._err(rc)
            self.close()
            raise hip.SvoError("svo_batch_create failed: " + msg)
        self.ctxs = [hip.Context.from_handle(self.L.svo_batch_context(self.h, k), self.Bc, max_kps) for k in range(contexts)]
        self._ck(self.L.svo_batch_set_params(self.h, C.byref(params)), "svo_batch_set_params")
        self._ck(self.L.svo_batch_set_camera(self.h, -1, C.byref(cam)), "svo_batch_set_camera")
        # the result records of every step land here (device memory the caller owns), in lane order
        self.rec = torch.zeros((lanes, C.sizeof(Result)), dtype=torch.uint8, device=self.dev)
        torch.cuda.synchronize(self.dev)                 # torch fills it on ITS stream: through before a context's stream writes records into it
        self._ck(self.L.svo_batch_set_results_buffer(self.h, C.c_void_p(self.rec.data_ptr()), C.c_size_t(self.rec.numel())), "svo_batch_set_results_buffer")

    def _err(self, rc):
        s = self.L.svo_strerror(rc).decode()
        if self.h:
            le = self.L.svo_batch_last_error(self.h)
            if le:
                s += " [" + le.decode() + "]"
        return s

    def _ck(self, rc, what):
        if rc < 0:
            raise hip.SvoError("%s: %s" % (what, self._err(rc)))
        return rc

    def step(self, ptrs, stride=None, pinned_host=False):
        """Enqueue one frame of every lane.  ptrs[lane] = (left, right) addresses of 8-bit grey images of the batch's
        size -- device memory, or page-locked host memory with pinned_host=True (the upload then runs on each context's
        copy stream and overlaps the kernels of the frames before it) -- lane = context * lanes_per_context +
        lane_in_context.  Returns at once; the result records land in self.rec (device) in lane order."""
        assert len(ptrs) == self.B
        fr = _frames(ptrs, self.W, self.H, self.W if stride is None else stride)
        self._ck(self.L.svo_batch_step(self.h, fr, C.c_uint32(hip.FLAG_PINNED_IMAGES if pinned_host else hip.FLAG_DEVICE_IMAGES)), "svo_batch_step")

    def prepare(self, ptrs, stride=None):
        """The frame table of one step built ahead of time (a few hundred ctypes structures: ~0.5 ms of Python per step otherwise, which
        is most of what the host spends on a step besides the launches themselves); hand the result to step_prepared."""
        assert len(ptrs) == self.B
        return _frames(ptrs, self.W, self.H, self.W if stride is None else stride)

    def step_prepared(self, fr, pinned_host=False):
        """step() with a frame table from prepare()."""
        self._ck(self.L.svo_batch_step(self.h, fr, C.c_uint32(hip.FLAG_PINNED_IMAGES if pinned_host else hip.FLAG_DEVICE_IMAGES)), "svo_batch_step")

    def flip_records(self):
        """Later steps leave their records in the OTHER of two buffers (self.rec afterwards): an all-gather may still be reading the
        one the last step wrote (svo_batch_switch_results_buffer: no synchronisation)."""
        if getattr(self, "_rec_alt", None) is None:
            self._rec_alt = torch.zeros_like(self.rec)
            torch.cuda.synchronize(self.dev)
        self.rec, self._rec_alt = self._rec_alt, self.rec
        self._ck(self.L.svo_batch_switch_results_buffer(self.h, C.c_void_p(self.rec.data_ptr()), C.c_size_t(self.rec.numel())), "svo_batch_switch_results_buffer")

    def make_wait(self, stream):
        """`stream` (torch) waits for the last step's work of every context (e.g. before an all-gather of self.rec)."""
        self._ck(self.L.svo_batch_wait_on_stream(self.h, C.c_void_p(stream.cuda_stream)), "svo_batch_wait_on_stream")

    def hold_for(self, event):
        """The next step's result copies wait for `event` (torch; e.g. the all-gather that still reads self.rec)."""
        self._ck(self.L.svo_batch_hold_for_event(self.h, C.c_void_p(event.cuda_event)), "svo_batch_hold_for_event")

    def synchronize(self):
        torch.cuda.synchronize(self.dev)
        self._ck(self.L.svo_batch_synchronize(self.h), "svo_batch_synchronize")

    def reset(self):
        """Every lane becomes a freshly constructed estimator again (common.cpp:28-50)."""
        torch.cuda.synchronize(self.dev)
        self._ck(self.L.svo_batch_reset(self.h), "svo_batch_reset")

    def lane(self, g):
        """(context, lane inside it) of global lane g."""
        return self.ctxs[g // self.Bc], g % self.Bc

    def results(self):
        arr = (Result * self.B)()
        self._ck(self.L.svo_batch_results(self.h, arr), "svo_batch_results")
        return list(arr)

    def pooled_kernel_times(self):
        """launches of all contexts pooled: ms and launch counts add up, a launch covers lanes_per_context streams"""
        acc = {}
        for c in self.ctxs:
            for name, v in c.kernel_times().items():
                a = acc.setdefault(name, [0.0, 0]); a[0] += v[0]; a[1] += v[1]
        return acc

    def close(self):
        if getattr(self, "h", None):
            self.L.svo_batch_destroy(self.h)
            self.h = None
        self.ctxs = []

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class FrameParallelStream:
    """ONE stereo stream (or one batch of `lanes` streams advancing together) whose consecutive frames are dealt
    round-robin to `contexts` contexts (SURVEY.md 8e "Within ONE stream"; svo_fpstream in include/svo_batch.h).  Same
    results as one context fed sequentially, bit for bit; frames per second are bounded by max(stages 4-5 of one frame,
    a whole frame / G) instead of a whole frame.

    On several GPUs the same protocol runs with one context per rank and the record sent rank g -> rank g + 1 with
    torch.distributed send / recv (RCCL over xGMI): see tools/frame_parallel_ranks.py."""

    def __init__(self, params, cam, width, height, lanes=1, contexts=2, device=0, max_kps=4096, max_cand=None, max_octaves=1):
        self.L = hip.lib()
        self.W, self.H, self.G, self.lanes = width, height, contexts, lanes
        self.dev = torch.device("cuda", device)
        if max_cand is None:
            max_cand = (1 << 18) if width * height > 2000000 else (1 << 17)
        cfg = hip.Config()
        self.L.svo_config_defaults(C.byref(cfg))
        cfg.device, cfg.n_lanes, cfg.max_w, cfg.max_h, cfg.max_kps, cfg.max_cand, cfg.max_octaves = device, lanes, width, height, max_kps, max_cand, int(max_octaves)
        h = C.c_void_p()
        rc = self.L.svo_fpstream_create(C.byref(cfg), contexts, C.byref(h))
        self.h = h
        if rc != 0:
            msg = self.L.svo_strerror(rc).decode() + (" [" + self.L.svo_fpstream_last_error(h).decode() + "]" if h else "")
            self.close()
            raise hip.SvoError("svo_fpstream_create failed: " + msg)
        self.ctxs = [hip.Context.from_handle(self.L.svo_fpstream_context(self.h, k), lanes, max_kps) for k in range(contexts)]
        self._ck(self.L.svo_fpstream_set_params(self.h, C.byref(params)), "svo_fpstream_set_params")
        self._ck(self.L.svo_fpstream_set_camera(self.h, -1, C.byref(cam)), "svo_fpstream_set_camera")
        self.t = 0

    def _ck(self, rc, what):
        if rc < 0:
            le = self.L.svo_fpstream_last_error(self.h)
            raise hip.SvoError("%s: %s [%s]" % (what, self.L.svo_strerror(rc).decode(), le.decode() if le else ""))
        return rc

    def push(self, ptrs, stride=None):
        """Enqueue the next frame (ptrs[lane] = (left, right) device addresses).  Returns the context that owns it."""
        fr = _frames(ptrs, self.W, self.H, self.W if stride is None else stride)
        self._ck(self.L.svo_fpstream_push(self.h, fr, C.c_uint32(hip.FLAG_DEVICE_IMAGES)), "svo_fpstream_push")
        c = self.ctxs[self.t % self.G]
        self.t += 1
        return c

    def synchronize(self):
        torch.cuda.synchronize(self.dev)
        self._ck(self.L.svo_fpstream_synchronize(self.h), "svo_fpstream_synchronize")

    def last_owner(self):
        return self.ctxs[(self.t - 1) % self.G]

    def close(self):
        if getattr(self, "h", None):
            self.L.svo_fpstream_destroy(self.h)
            self.h = None
        self.ctxs = []

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
