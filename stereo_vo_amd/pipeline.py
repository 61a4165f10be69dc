"""Scheduling of several batched contexts on ONE GPU: the step that bench.py times and the GPU tests check.

A context (hip.Context) drives up to 64 independent estimator streams ("lanes") through every kernel launch.  This
module runs `contexts` of them side by side so that the per-stream, latency-bound kernels of stages 3-5 of one context
overlap the throughput kernels of stage 2 of the next:

  schedule "pipelined": ONE HIP stream carries the detect phases (stage 2) of all contexts back to back, ANOTHER carries
      stages 3-5 of each frame; the stage 3-5 stream has the higher priority by default (det_priority = "low").  Events: stages 3-5 of context k wait for detect(k);
      the next detect of context k waits for its stages 3-5 (stage 4 reads the feature slot detection overwrites next).
  schedule "free": every context runs its whole frame on its own stream, unsynchronised with the others.

Product path only: nothing here touches oracle/.  The reference has no counterpart (one estimator, one thread:
libstereo-odometry.h:732-831 is all per-instance state, which is what makes the lanes independent).
"""
import ctypes as C

import torch

from . import hip
from .abi import Result


class StreamBatch:
    def __init__(self, params, cam, width, height, lanes, contexts=1, device=0, schedule="pipelined", post_on_rest=False,
                 det_priority="low", kernel_times=False, max_octaves=1, max_kps=4096, max_cand=None, det_streams=1):
        assert contexts >= 1 and lanes % contexts == 0 and lanes // contexts <= hip.MAX_LANES, \
            "lanes must split evenly over the contexts, at most %d streams per context" % hip.MAX_LANES
        self.W, self.H, self.B, self.NC, self.Bc = width, height, lanes, contexts, lanes // contexts
        self.dev = torch.device("cuda", device)
        self.pipelined = contexts > 1 and schedule == "pipelined"
        # post_on_rest: False = NMS + describe stay on the detect stream; True = they run on the stage 3-5 stream; "own" = on a
        # third stream, so that the detect stream goes straight on to the next context's resize + FAST
        self.post_own = post_on_rest == "own"
        self.post_on_rest = bool(post_on_rest)
        if max_cand is None:
            max_cand = (1 << 18) if width * height > 2000000 else (1 << 17)
        self.streams = [torch.cuda.Stream(self.dev) for _ in range(contexts)]
        self.ctxs = []
        for k in range(contexts):
            c = hip.Context(n_lanes=self.Bc, max_w=width, max_h=height, max_kps=max_kps, device=device, kernel_times=kernel_times,
                            stream=self.streams[k].cuda_stream, max_octaves=max_octaves, max_cand=max_cand)
            c.set_params(params); c.set_camera(cam)
            self.ctxs.append(c)
        self.rec = torch.zeros((lanes, C.sizeof(Result)), dtype=torch.uint8, device=self.dev)
        # det_streams > 1: the detect phases of consecutive contexts go to different streams, so that the latency-bound tail of
        # one context's stage 2 (small pyramid levels, selection, NMS) overlaps the throughput kernels of the next one's
        self.s_dets = [torch.cuda.Stream(self.dev, priority=-1 if det_priority == "high" else 0) for _ in range(max(1, det_streams))]
        self.s_det = self.s_dets[0]
        self.s_rest = torch.cuda.Stream(self.dev, priority=0 if det_priority == "high" else -1)
        self.det_done = [torch.cuda.Event() for _ in range(contexts)]
        self.rest_done = [torch.cuda.Event() for _ in range(contexts)]
        self.done = [torch.cuda.Event() for _ in range(contexts)]
        self.first = True
        self.REST = hip.RUN_MATCH | hip.RUN_TRACK | hip.RUN_OPTIMIZE | (hip.RUN_DETECT_POST if (post_on_rest and not self.post_own) else 0)
        self.s_post = torch.cuda.Stream(self.dev, priority=0 if det_priority == "high" else -1) if self.post_own else None
        self.pre_done = [torch.cuda.Event() for _ in range(contexts)]

    def step(self, ptrs, stride=None, pinned_host=False):
        """Enqueue one frame of every lane.  ptrs[lane] = (left, right) addresses of 8-bit grey images of the batch's
        size -- device memory, or page-locked host memory with pinned_host=True (the upload then runs on each context's
        copy stream and overlaps the kernels of the frames before it) -- lane = context * lanes_per_context +
        lane_in_context.  Returns at once; the result records land in self.rec (device) in lane order."""
        assert len(ptrs) == self.B
        stride = self.W if stride is None else stride
        Bc, rsz = self.Bc, C.sizeof(Result)
        for k, c in enumerate(self.ctxs):
            pk = ptrs[k * Bc:(k + 1) * Bc]
            proc = c.process_pinned if pinned_host else c.process_device
            if self.pipelined:
                s_det = self.s_dets[k % len(self.s_dets)]
                if not self.first:
                    s_det.wait_event(self.rest_done[k])
                c.set_stream(s_det.cuda_stream)
                proc(pk, self.W, self.H, stride, hip.RUN_DETECT | (hip.FLAG_DETECT_NO_POST if self.post_on_rest else 0))
                if self.post_own:
                    self.pre_done[k].record(s_det)
                    self.s_post.wait_event(self.pre_done[k])
                    c.set_stream(self.s_post.cuda_stream)
                    c.run_stages(hip.RUN_DETECT_POST)
                    self.det_done[k].record(self.s_post)
                else:
                    self.det_done[k].record(s_det)
                self.s_rest.wait_event(self.det_done[k])
                c.set_stream(self.s_rest.cuda_stream)
                c.run_stages(self.REST)
                c.copy_results_async(self.rec[k * Bc:(k + 1) * Bc].data_ptr(), Bc * rsz)
                self.rest_done[k].record(self.s_rest)
            else:
                proc(pk, self.W, self.H, stride)
                c.copy_results_async(self.rec[k * Bc:(k + 1) * Bc].data_ptr(), Bc * rsz)
                self.done[k].record(self.streams[k])
        self.first = False

    def make_wait(self, stream):
        """`stream` (torch) waits for the last step's work of every context (e.g. before an all-gather of self.rec)."""
        for k in range(self.NC):
            stream.wait_event(self.rest_done[k] if self.pipelined else self.done[k])

    def hold_for(self, event):
        """The next step's result copies wait for `event` (e.g. the all-gather that still reads self.rec)."""
        if self.pipelined:
            self.s_rest.wait_event(event)
        else:
            for s in self.streams:
                s.wait_event(event)

    def synchronize(self):
        torch.cuda.synchronize(self.dev)
        for c in self.ctxs:
            c.wait()

    def reset(self):
        """Every lane becomes a freshly constructed estimator again (common.cpp:28-50)."""
        self.synchronize()
        for c in self.ctxs:
            c.set_stream(None)
            c.reset(-1)
        self.first = True

    def lane(self, g):
        """(context, lane inside it) of global lane g."""
        return self.ctxs[g // self.Bc], g % self.Bc

    def results(self):
        out = []
        for c in self.ctxs:
            out += c.results()
        return out

    def pooled_kernel_times(self):
        """launches of all contexts pooled: ms and launch counts add up, a launch covers lanes_per_context streams"""
        acc = {}
        for c in self.ctxs:
            for name, v in c.kernel_times().items():
                a = acc.setdefault(name, [0.0, 0]); a[0] += v[0]; a[1] += v[1]
        return acc

    def close(self):
        for c in self.ctxs:
            c.close()
        self.ctxs = []


class FrameParallelStream:
    """ONE stereo stream (or one batch of `lanes` streams advancing together) whose consecutive frames are dealt
    round-robin to `contexts` contexts (SURVEY.md 8e "Within ONE stream").  Context g = t % G runs stages 2-3 of frame t
    on its own HIP stream as soon as it is free -- overlapping stages 2-5 of frame t-1 on the context before it -- then
    imports the previous owner's hand-over record (svo_export_frame / svo_import_frame), runs stages 4-5 and exports its
    own.  Same results as one context fed sequentially, bit for bit; frames per second are bounded by
    max(stages 4-5 of one frame, a whole frame / G) instead of a whole frame.

    On several GPUs the same protocol runs with one context per rank and the record sent rank g -> rank g + 1 with
    torch.distributed send / recv (RCCL over xGMI): see tools/frame_parallel_ranks.py."""

    def __init__(self, params, cam, width, height, lanes=1, contexts=2, device=0, max_kps=4096, max_cand=None, max_octaves=1):
        self.W, self.H, self.G, self.lanes = width, height, contexts, lanes
        self.dev = torch.device("cuda", device)
        if max_cand is None:
            max_cand = (1 << 18) if width * height > 2000000 else (1 << 17)
        self.streams = [torch.cuda.Stream(self.dev) for _ in range(contexts)]
        self.ctxs = []
        for g in range(contexts):
            c = hip.Context(n_lanes=lanes, max_w=width, max_h=height, max_kps=max_kps, device=device, stream=self.streams[g].cuda_stream,
                            max_octaves=max_octaves, max_cand=max_cand)
            c.set_params(params); c.set_camera(cam)
            self.ctxs.append(c)
        self.nbytes = self.ctxs[0].handover_bytes()
        self.blobs = [torch.zeros(self.nbytes, dtype=torch.uint8, device=self.dev) for _ in range(contexts)]
        self.exported = [torch.cuda.Event() for _ in range(contexts)]
        self.rec = torch.zeros((contexts, lanes, C.sizeof(Result)), dtype=torch.uint8, device=self.dev)
        self.t = 0

    def push(self, ptrs, stride=None):
        """Enqueue the next frame (ptrs[lane] = (left, right) device addresses).  Returns the context that owns it."""
        stride = self.W if stride is None else stride
        g, t = self.t % self.G, self.t
        c, s = self.ctxs[g], self.streams[g]
        c.process_device(ptrs, self.W, self.H, stride, hip.RUN_DETECT | hip.RUN_MATCH)            # stages 2-3: independent of every other frame
        if t > 0:
            gp = (t - 1) % self.G
            s.wait_event(self.exported[gp])
            c.import_frame(self.blobs[gp].data_ptr(), self.nbytes)
        c.run_stages(hip.RUN_TRACK | hip.RUN_OPTIMIZE)
        c.export_frame(self.blobs[g].data_ptr(), self.nbytes)
        self.exported[g].record(s)
        c.copy_results_async(self.rec[g].data_ptr(), self.lanes * C.sizeof(Result))
        self.t += 1
        return c

    def synchronize(self):
        torch.cuda.synchronize(self.dev)
        for c in self.ctxs:
            c.wait()

    def last_owner(self):
        return self.ctxs[(self.t - 1) % self.G]

    def close(self):
        for c in self.ctxs:
            c.close()
        self.ctxs = []
