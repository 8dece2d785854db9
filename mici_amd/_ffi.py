"""ctypes binding of libmici_amd.so (include/mici_amd.h).  There is no fallback: if the shared
library is missing or a call fails, a DeviceError is raised."""

from __future__ import annotations

import ctypes as C
import os

from .errors import DeviceError

# MICI_AMD_LIB: another build of the library (A/B runs of kernel variants, tools/ab_build.py)
_LIB_PATH = os.environ.get("MICI_AMD_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libmici_amd.so")

MM_COMM_ID_BYTES = 128
ABI_VERSION = 4

c_double_p = C.POINTER(C.c_double)
c_int8_p = C.POINTER(C.c_int8)
c_int32_p = C.POINTER(C.c_int32)
c_uint8_p = C.POINTER(C.c_uint8)


class ModelDesc(C.Structure):
    _fields_ = [
        ("dim", C.c_int32),
        ("target", C.c_int32),
        ("target_params", c_double_p),
        ("n_target_params", C.c_size_t),
        ("metric_kind", C.c_int32),
        ("gaussian_split", C.c_int32),
        ("metric", c_double_p),
        ("n_metric", C.c_size_t),
        ("rmetric", C.c_int32),
        ("n_constr", C.c_int32),
        ("rmetric_params", c_double_p),
        ("n_rmetric_params", C.c_size_t),
        ("constr", C.c_int32),
        ("dens_wrt_ambient", C.c_int32),
        ("constr_params", c_double_p),
        ("n_constr_params", C.c_size_t),
    ]


class FpOpts(C.Structure):
    _fields_ = [
        ("conv_tol", C.c_double),
        ("div_tol", C.c_double),
        ("max_iters", C.c_int32),
        ("norm", C.c_int32),
        ("solver", C.c_int32),
        ("rev_norm", C.c_int32),
        ("rev_tol", C.c_double),
    ]


class ProjOpts(C.Structure):
    _fields_ = [
        ("constr_tol", C.c_double),
        ("pos_tol", C.c_double),
        ("div_tol", C.c_double),
        ("max_iters", C.c_int32),
        ("norm", C.c_int32),
        ("solver", C.c_int32),
        ("rev_norm", C.c_int32),
        ("rev_tol", C.c_double),
        ("n_inner", C.c_int32),
        ("max_line_search_iters", C.c_int32),
    ]


class Counters(C.Structure):
    _fields_ = [(n, C.c_int64) for n in (
        "n_grad", "n_metric", "n_inverse", "n_fp_evals", "n_fp_solves", "n_newton_iters",
        "n_constr", "n_eigh", "n_refine", "n_factor_full", "n_factor_solve", "n_mfma_products", "n_lowrank", "n_inverse_update")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


# every symbol include/mici_amd.h declares: name -> (restype, argtypes)
_VP = C.c_void_p
SIGNATURES = {
    "mm_abi_version": (C.c_int, []),
    "mm_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "mm_ctx_create": (C.c_int, [C.c_int, C.POINTER(_VP)]),
    "mm_ctx_destroy": (C.c_int, [_VP]),
    "mm_ctx_sync": (C.c_int, [_VP]),
    "mm_last_error": (C.c_char_p, [_VP]),
    "mm_ctx_record": (C.c_int, [_VP, C.c_int]),
    "mm_ctx_elapsed_ms": (C.c_int, [_VP, C.c_int, C.c_int, c_double_p]),
    "mm_model_create": (C.c_int, [_VP, C.POINTER(ModelDesc), C.POINTER(_VP)]),
    "mm_model_create_from_source": (C.c_int, [_VP, C.POINTER(ModelDesc), C.c_char_p, C.POINTER(_VP)]),
    "mm_model_destroy": (C.c_int, [_VP]),
    "mm_state_alloc": (C.c_int, [_VP, C.c_int64, C.c_int32, C.POINTER(_VP)]),
    "mm_state_alloc_mapped": (C.c_int, [_VP, C.c_int64, C.c_int32, C.POINTER(_VP)]),
    "mm_state_free": (C.c_int, [_VP]),
    "mm_state_upload": (C.c_int, [_VP, c_double_p, c_double_p, c_int8_p]),
    "mm_state_download": (C.c_int, [_VP, c_double_p, c_double_p, c_int8_p]),
    "mm_state_download_status": (C.c_int, [_VP, c_int32_p, c_int32_p]),
    "mm_state_download_all": (C.c_int, [_VP, c_double_p, c_double_p, c_int8_p, c_int32_p, c_int32_p]),
    "mm_state_device_ptrs": (C.c_int, [_VP, C.POINTER(_VP), C.POINTER(_VP), C.POINTER(_VP)]),
    "mm_momentum_refresh": (C.c_int, [_VP, _VP, _VP, c_double_p, C.c_double]),
    "mm_state_copy": (C.c_int, [_VP, _VP]),
    "mm_state_set_step_scale": (C.c_int, [_VP, c_double_p]),
    "mm_state_set_chain_steps": (C.c_int, [_VP, c_int32_p]),
    "mm_metropolis_accept": (C.c_int, [_VP, _VP, _VP, _VP, c_double_p, c_double_p, c_int8_p]),
    "mm_state_set_rng": (C.c_int, [_VP, C.c_uint64, C.c_uint64]),
    "mm_momentum_refresh_rng": (C.c_int, [_VP, _VP, _VP, C.c_double, C.c_uint64]),
    "mm_metropolis_accept_rng": (C.c_int, [_VP, _VP, _VP, _VP, C.c_uint64, c_double_p, c_int8_p]),
    "mm_rng_chain_steps": (C.c_int, [_VP, C.c_uint64, C.c_int32, C.c_int32]),
    "mm_rng_draws": (C.c_int, [_VP, C.c_uint64, c_double_p, c_double_p, c_int32_p, C.c_int32, C.c_int32]),
    "mm_leapfrog_euclid": (C.c_int, [_VP, _VP, _VP, C.c_double, C.c_int32]),
    "mm_composition_euclid": (C.c_int, [_VP, _VP, _VP, C.c_double, C.c_int32, C.c_int32, c_double_p, C.c_int32]),
    "mm_implicit_leapfrog": (C.c_int, [_VP, _VP, _VP, C.c_double, C.c_int32,
                                       C.POINTER(FpOpts), C.POINTER(Counters)]),
    "mm_implicit_midpoint": (C.c_int, [_VP, _VP, _VP, C.c_double, C.c_int32,
                                       C.POINTER(FpOpts), C.POINTER(Counters)]),
    "mm_constrained_leapfrog": (C.c_int, [_VP, _VP, _VP, C.c_double, C.c_int32,
                                          C.POINTER(ProjOpts), C.POINTER(Counters)]),
    "mm_hamiltonian": (C.c_int, [_VP, _VP, _VP, c_double_p]),
    "mm_dh_dmom": (C.c_int, [_VP, _VP, _VP, c_double_p]),
    "mm_sample_momentum": (C.c_int, [_VP, _VP, _VP, c_double_p]),
    "mm_state_download_errors": (C.c_int, [_VP, C.POINTER(C.c_uint32), C.c_int32]),
    "mm_state_mapped_ptrs": (C.c_int, [_VP, C.POINTER(_VP), C.POINTER(_VP), C.POINTER(_VP), C.POINTER(_VP),
                                       C.POINTER(_VP)]),
    "mm_comm_unique_id": (C.c_int, [c_uint8_p]),
    "mm_comm_create": (C.c_int, [_VP, C.c_int32, C.c_int32, c_uint8_p, C.POINTER(_VP)]),
    "mm_comm_destroy": (C.c_int, [_VP]),
    "mm_comm_allgather_pos": (C.c_int, [_VP, _VP, c_double_p]),
    "mm_comm_allgather_pos_async": (C.c_int, [_VP, _VP, C.c_int]),
    "mm_comm_wait": (C.c_int, [_VP, c_double_p]),
    "mm_comm_count": (C.c_int, [_VP, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
}

_DEV_LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libmici_amd_dev.so")
_libs = {}


def lib_path(dev=False):
    return _DEV_LIB_PATH if dev else _LIB_PATH


def load(dev=False):
    """Load libmici_amd.so and bind every declared symbol (no GPU needed for this).  dev=True: the developer build
    libmici_amd_dev.so - the same library plus the test / micro-benchmark kernels (mici_amd/build.py) - as a second,
    independent copy: objects created through a `Context(dev=True)` stay on it."""
    if dev in _libs:
        return _libs[dev]
    path = lib_path(dev)
    if not os.path.exists(path):
        raise DeviceError(
            f"{path} is missing: build it with `python -m mici_amd.build` "
            "(hipcc, --offload-arch=gfx950). mici_amd has no CPU fallback."
        )
    try:
        lib = C.CDLL(path, mode=C.RTLD_LOCAL if dev else C.RTLD_GLOBAL)
    except OSError as e:
        raise DeviceError(f"cannot load {path}: {e}") from e
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise DeviceError(f"{path} does not export {name}") from e
        fn.restype = res
        fn.argtypes = args
    if lib.mm_abi_version() != ABI_VERSION:
        raise DeviceError(f"{os.path.basename(path)} ABI version mismatch; rebuild with mici_amd.build")
    _libs[dev] = lib
    return lib


def check(rc, ctx=None, what=""):
    if rc == 0:
        return
    msg = load().mm_last_error(ctx)
    msg = msg.decode("utf-8", "replace") if msg else ""
    raise DeviceError(f"{what or 'libmici_amd call'} failed (rc={rc}): {msg}")
