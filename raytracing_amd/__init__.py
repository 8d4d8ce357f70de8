"""raytracing_amd -- MI355X (gfx950) implementation of the per-pixel hot path of the
lukeovalle/raytracing spectral path tracer, behind a C ABI (include/rtmi.h).

csrc/            HIP kernels + host runtime, built into librtmi.so
_lib.py, api.py  ctypes binding and thin handles (no CPU fallback)
scene_builder.py the reference's loader semantics for building flat scenes
scenes.py        the benchmark / parity workloads
parallel.py      sample-range and tile partitioning across ranks (torch.distributed over RCCL)
"""
from ._lib import LIB_PATH, RtmiError, build  # noqa: F401
