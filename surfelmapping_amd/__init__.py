"""surfelmapping_amd -- MI355X-native surfel-fusion core behind the SurfelMapping API.

Only what the per-frame hot path needs lives here:
  csrc/     hand-written HIP kernels for gfx950 + the C-ABI (include/sm_c_api.h) + C++ facade
  capi.py   ctypes binding of the C-ABI (host-side mirror of SurfelMapping/GlobalModel/IndexMap)
  synth.py  deterministic synthetic RGB-D+semantic frames (no dataset is available offline)

There is no CPU fallback: the compute path is the HIP library or an error.
"""
from . import capi, synth  # noqa: F401
from .capi import SurfelMap, SurfelMapError, make_config  # noqa: F401

__all__ = ["capi", "synth", "SurfelMap", "SurfelMapError", "make_config"]
