"""Optional roctx ranges around the phases of the launch schedules (FS2_TRACE=1), so that
`rocprofv3 --marker-trace --kernel-trace` groups kernels by phase (phone level / frame level / FFT stack /
backward).  The reference has no tracing at all; without FS2_TRACE, or without libroctx64,
`span()` is a no-op with no per-call cost beyond one attribute test."""
import contextlib
import ctypes
import os

_push = _pop = None
if os.environ.get("FS2_TRACE"):
    for _name in ("libroctx64.so", "/opt/rocm/lib/libroctx64.so", "librocprofiler-sdk-roctx.so",
                  "/opt/rocm/lib/librocprofiler-sdk-roctx.so"):
        try:
            _lib = ctypes.CDLL(_name)
            _push, _pop = _lib.roctxRangePushA, _lib.roctxRangePop
            _push.argtypes = [ctypes.c_char_p]
            break
        except (OSError, AttributeError):
            continue

ENABLED = _push is not None


@contextlib.contextmanager
def _span(name: str):
    _push(name.encode())
    try:
        yield
    finally:
        _pop()


_NULL = contextlib.nullcontext()


def span(name: str):
    """`with span("frame_phase"): ...` — a roctx range when tracing is on, else nothing."""
    return _span(name) if ENABLED else _NULL
