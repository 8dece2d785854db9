"""mici_amd: the symplectic-integrator hot path of matt-graham/mici on AMD MI355X (gfx950).

Hand-written HIP kernels (``csrc/``) behind a C ABI (``include/mici_amd.h``, ``lib/libmici_amd.so``)
and a Python host mirror of the reference's System / Integrator surface.  No CPU fallback."""

from . import adapters, errors, integrators, models, solvers, states, systems, traces, transitions  # noqa: F401
from .runtime import Context, DeviceBatch, default_context  # noqa: F401
from .states import ChainState  # noqa: F401

__all__ = ["adapters", "errors", "integrators", "models", "solvers", "states", "systems", "traces",
           "transitions", "Context",
           "DeviceBatch", "default_context", "ChainState"]
