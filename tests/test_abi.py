"""CPU-side checks of the C-ABI library: it loads, exports every symbol include/svo_hip.h declares, its record
sizes match the Python mirrors, and it refuses to run without a HIP device (no CPU fallback)."""
import ctypes as C
import os
import re
import pytest

from stereo_vo_amd import hip
from stereo_vo_amd.abi import Params, Result, StereoCamera, keypoint_dtype, dmatch_dtype

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    L = hip.lib()
    hdr = open(os.path.join(ROOT, "include", "svo_hip.h")).read()
    declared = set(re.findall(r"\b(svo_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(hip.EXPORTS), declared ^ set(hip.EXPORTS)
    for name in declared:
        assert getattr(L, name) is not None


def test_ab_kernel_forms_live_in_their_own_library():
    """the A/B-only kernel forms (csrc/svo_kernels.h: the int8 matcher, two RANSAC count forms) are compiled under
    -DSVO_AB_KERNELS into libsvo_hip_ab.so only; the product library neither defines nor launches them.  Both export the same C-ABI."""
    import subprocess
    prod, ab = os.path.join(ROOT, "stereo_vo_amd", "libsvo_hip.so"), os.path.join(ROOT, "stereo_vo_amd", "libsvo_hip_ab.so")
    assert os.path.exists(ab), "libsvo_hip_ab.so missing: run __graft_entry__.build()"
    syms = {p: subprocess.run(["nm", "-D", "--defined-only", p], capture_output=True, text=True).stdout for p in (prod, ab)}
    for kernel in ("k_gather_mdesc", "19k_ransac_count_mfma6DevCtxi", "9k_hamming6DevCtxii"):
        assert kernel not in syms[prod] and kernel in syms[ab], kernel
    for kernel in ("k_hamming_f4", "k_ransac_count_mfma16", "k_ransac_hyp_thread"):
        assert kernel in syms[prod] and kernel in syms[ab], kernel
    Lab = C.CDLL(ab)
    for name in hip.EXPORTS + hip.BATCH_EXPORTS:
        assert getattr(Lab, name) is not None


def test_batch_scheduler_exports_every_declared_symbol():
    """include/svo_batch.h: the batched / pipelined and frame-parallel schedulers live in libsvo_hip.so, behind the C-ABI"""
    L = hip.lib()
    hdr = open(os.path.join(ROOT, "include", "svo_batch.h")).read()
    declared = set(re.findall(r"\b(svo_(?:batch|fpstream)_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(hip.BATCH_EXPORTS), declared ^ set(hip.BATCH_EXPORTS)
    for name in declared:
        assert getattr(L, name) is not None
    assert L.svo_batch_lanes(None) < 0 and L.svo_batch_create(None, None) < 0 and L.svo_fpstream_push(None, None, 0) < 0
    assert L.svo_batch_switch_results_buffer(None, None, 0) < 0 and L.svo_batch_set_results_buffer(None, None, 0) < 0


def test_rccl_companion_exports_every_declared_symbol():
    """include/svo_rccl.h: the exchange steps of the multi-GPU path; loading it needs no GPU, creating a group does"""
    from stereo_vo_amd import rccl
    L = rccl.lib()
    hdr = open(os.path.join(ROOT, "include", "svo_rccl.h")).read()
    declared = set(re.findall(r"\b(svo_group_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(rccl.EXPORTS), declared ^ set(rccl.EXPORTS)
    for name in declared:
        assert getattr(L, name) is not None
    assert L.svo_group_size(None) < 0 and L.svo_group_create_local(None, 0, None) < 0


def test_abi_record_sizes():
    a = (C.c_int32 * 6)()
    hip.lib().svo_abi_sizes(a)
    assert list(a) == [keypoint_dtype.itemsize, dmatch_dtype.itemsize, C.sizeof(StereoCamera), C.sizeof(Params), C.sizeof(Result), C.sizeof(hip.Config)]


def test_defaults_agree_with_oracle_defaults():
    from oracle import oracle as O
    a, b = hip.default_params(), O.default_params()
    assert bytes(a) == bytes(b)


def test_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(hip.SvoError, match="no HIP device"):
        hip.Context()


def test_strerror():
    assert hip.lib().svo_strerror(0) == b"ok" and b"fallback" in hip.lib().svo_strerror(-4)


def test_batch_config_defaults_are_the_measured_schedule():
    """svo_batch_config_defaults: three contexts, pipelined, NMS + describe + stages 3-5 on one stream per context (post_mode 1,
    rest_streams 0), one detect stream at high priority -- the schedule bench.py times (include/svo_batch.h)"""
    from stereo_vo_amd import hip
    cfg = hip.BatchConfig()
    hip.lib().svo_batch_config_defaults(C.byref(cfg))
    assert (cfg.n_contexts, cfg.schedule, cfg.det_priority_high, cfg.post_mode, cfg.det_streams, cfg.rest_streams, cfg.no_detect_ahead) == (2, 0, 1, 1, 1, 0, 0)
    assert cfg.ctx.n_lanes == 96


def test_batch_config_size_is_checked_at_the_boundary():
    """svo_batch_config grew a trailing field and SVO_MAX_LANES doubled without a guard.  The library now reports its own
    sizes (svo_batch_abi_sizes), the header's svo_batch_create() macro passes the caller's sizeof along, and a mismatch is refused
    with both numbers in svo_batch_last_error -- before any device is touched."""
    from stereo_vo_amd import hip
    L = hip.lib()
    out = (C.c_int32 * 3)()
    L.svo_batch_abi_sizes(out)
    hdr = open(os.path.join(ROOT, "include", "svo_batch.h")).read()
    assert out[0] == C.sizeof(hip.BatchConfig) and out[2] == int(re.search(r"#define SVO_BATCH_ABI_VERSION (\d+)", hdr).group(1))
    assert out[1] == int(re.search(r"#define SVO_MAX_LANES (\d+)", open(os.path.join(ROOT, "include", "svo_hip.h")).read()).group(1))
    cfg = hip.BatchConfig()
    L.svo_batch_config_defaults(C.byref(cfg))
    h = C.c_void_p()
    rc = L.svo_batch_create_sized(C.byref(cfg), C.sizeof(cfg) - 4, C.byref(h))          # a host built against the round-3 header
    assert rc == -2 and h.value
    msg = L.svo_batch_last_error(h).decode()
    assert str(C.sizeof(cfg) - 4) in msg and str(C.sizeof(cfg)) in msg and "ABI version" in msg
    L.svo_batch_destroy(h)
