"""Kernel-name matching shared by the profile summarisers: rocprofv3 prints templated kernels as
`void k_select<2048>(DevCtx, int)`, plain ones as `k_fast(DevCtx)` -- both are this repo's kernels."""
import re

_K = re.compile(r"\bk_\w+")


def kname(full):
    """`k_xxx` of a rocprofv3 kernel name, or None when the kernel is not one of this repo's."""
    m = _K.search(full)
    return m.group(0) if m else None
