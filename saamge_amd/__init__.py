"""saamge_amd -- MI355X-native (gfx950, HIP) implementation of the SAAMGE setup+solve hot
path behind a C ABI (include/saamge_amd.h).  The Python side is only the test / bench
harness: `capi` marshals arrays through the C ABI, `problems` generates synthetic inputs."""
from . import problems  # noqa: F401

__all__ = ["problems", "capi"]
