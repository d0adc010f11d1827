"""ctypes binding of libmbvae_hip.so (the C ABI declared in include/mbvae_hip.h).

The product path has no CPU or eager-PyTorch fallback: if the shared library is missing, or a kernel
entry point returns a non-zero status, a RuntimeError is raised.
"""
import ctypes
import os
from ctypes import c_char_p, c_float, c_int, c_int64, c_uint64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
# MBVAE_LIB selects another build of the same library (kernel A/B timing); the default is the in-tree build.
LIB_PATH = os.environ.get("MBVAE_LIB") or os.path.join(os.path.dirname(_HERE), "libmbvae_hip.so")

_P, _I, _L, _F, _U = c_void_p, c_int, c_int64, c_float, c_uint64

# name -> argtypes (restype is int unless listed in _RET64)
SIGNATURES = {
    "mbvae_set_option": [c_char_p, _I],
    "mbvae_get_option": [c_char_p],
    "mbvae_linear_fwd": [_P, _L, _P, _P, _P, _L, _L, _L, _L, _I, _P, _I, _I, _F, _U, _I, _P, _L, _P],
    "mbvae_linear_bwd_input": [_P, _L, _P, _P, _L, _L, _L, _L, _P, _F, _P, _I, _I, _I, _P, _L, _P],
    "mbvae_cycle_counts": [_P, _P, _P, _L, _L, _P],
    "mbvae_linear_fwd_bf16": [_P, _L, _P, _P, _P, _L, _L, _L, _L, _P, _I, _I, _P],
    "mbvae_linear_bwd_input_bf16": [_P, _I, _L, _P, _P, _P, _I, _L, _L, _L, _L, _P, _I, _I, _P],
    "mbvae_linear_bwd_weight_bf16": [_P, _I, _L, _P, _I, _L, _P, _P, _L, _L, _L, _P, _I, _I, _P, _L, _P],
    "mbvae_linear_bwd_input_add": [_P, _L, _P, _P, _P, _L, _L, _L, _L, _P, _F, _P, _I, _I, _P],
    "mbvae_linear_bwd_weight": [_P, _L, _P, _L, _P, _P, _L, _L, _L, _P, _F, _P, _I, _I, _I, _P, _L, _P],
    "mbvae_linear_bwd_input_outgate": [_P, _L, _P, _P, _L, _L, _L, _L, _P, _F, _P, _I, _I, _P, _L, _P],
    "mbvae_film_fwd": [_P, _P, _P, _P, _L, _L, _L, _P],
    "mbvae_film_bwd": [_P, _P, _P, _P, _P, _P, _L, _L, _L, _P],
    "mbvae_gate_fwd": [_P, _P, _P, _P, _F, _P, _L, _P],
    "mbvae_gate_bwd": [_P, _P, _P, _P, _F, _P, _P, _L, _I, _P],
    "mbvae_linear_bwd_both": [_P, _L, _P, _P, _L, _P, _P, _L, _P, _P, _L, _L, _L, _P, _I, _I, _P, _L, _P],
    "mbvae_edge_attn_fwd": [_P, _P, _P, _L, _P, _P, _L] + [_P] * 6 + [_L, _L, _L, _F, _P],
    "mbvae_edge_attn_bwd_workspace": [_L, _L, _L],
    "mbvae_edge_attn_bwd": [_P, _P, _P, _L, _P, _P, _L] + [_P] * 13 + [_L, _L, _L, _F, _P, _L, _P],
    "mbvae_edge_fused_fwd": [_P] * 6 + [_L] + [_P] * 10 + [_L, _L, _L, _L, _F, _P],
    "mbvae_edge_attn_bwd_bf16": [_P, _P, _P, _L, _P, _P, _L] + [_P] * 13 + [_L, _L, _L, _F, _P, _L, _P],
    "mbvae_edge_fused_fwd_recompute": [_P] * 6 + [_L] + [_P] * 6 + [_L, _L, _L, _L, _F, _P],
    "mbvae_edge_fused_bwd": [_P] * 6 + [_L] + [_P] * 12 + [_L, _L, _L, _L, _F, _P],
    "mbvae_edge_fused_fwd_recompute_bf16": [_P] * 6 + [_L] + [_P] * 6 + [_L, _L, _L, _L, _F, _P],
    "mbvae_edge_fused_bwd_bf16": [_P] * 6 + [_L] + [_P] * 12 + [_L, _L, _L, _L, _F, _P],
    "mbvae_edge_fused_fwd_bf16": [_P] * 6 + [_L] + [_P] * 7 + [_L, _L, _L, _L, _F, _P],
    "mbvae_edge_ffn_fwd": [_P] * 11 + [_L, _L, _L, _F, _P],
    "mbvae_edge_ffn_fwd_train": [_P] * 10 + [_L, _L, _L, _F, _F, _U, _F, _U, _P],
    "mbvae_edge_ffn_bwd_data": [_P] * 7 + [_L, _L, _L, _F, _P],
    "mbvae_edge_out_ln_fwd": [_P] * 5 + [_L] + [_P] * 6 + [_I, _L, _L, _L, _F, _P],
    "mbvae_ln_residual_fwd": [_P] * 7 + [_L, _L, _F, _F, _U, _P],
    "mbvae_linear_ln_fwd": [_P, _L] + [_P] * 7 + [_L, _L, _L, _P, _I, _I, _F, _F, _U, _P],
    "mbvae_linear_ln_fwd_bf16": [_P, _L] + [_P] * 7 + [_L, _L, _L, _P, _I, _I, _F, _F, _U, _P],
    "mbvae_ln_sum_bwd": [_P] * 7 + [_L, _L, _F, _F, _U, _P, _L, _P],
    "mbvae_ln_residual_bwd_workspace": [_L],
    "mbvae_ln_residual_bwd": [_P] * 10 + [_L, _L, _F, _U, _P, _L, _P],
    "mbvae_pool_stats_fwd": [_P, _P, _P, _L, _L, _L, _P],
    "mbvae_pool_stats_nodes_fwd": [_P, _P, _P, _L, _L, _L, _L, _P],
    "mbvae_pool_stats_pairs_fwd": [_P, _P, _P, _P, _P, _L, _L, _L, _P],
    "mbvae_pool_stats_bwd": [_P, _P, _P, _P, _P, _L, _L, _L, _I, _P],
    "mbvae_sym_mask": [_P, _P, _L, _P, _P, _L, _L, _L, _I, _P],
    "mbvae_masked_ce_workspace": [],
    "mbvae_masked_ce_fwd": [_P, _P, _P, _L, _L, _P, _L, _P],
    "mbvae_masked_ce_fwd_static": [_P, _P, _P, _L, _L, _P],
    "mbvae_masked_ce_bwd": [_P, _P, _P, _P, _L, _L, _P],
    "mbvae_kl_workspace": [],
    "mbvae_kl_reparam_fwd": [_P, _P, _P, _P, _P, _L, _P, _L, _P],
    "mbvae_kl_reparam_bwd": [_P, _P, _P, _P, _P, _P, _P, _L, _P],
    "mbvae_sym_eig": [_P, _P, _P, _L, _L, _P],
    "mbvae_dense_from_ids": [_P] * 5 + [_L, _L, _L, _L, _P],
    "mbvae_apply_noise": [_P] * 12 + [_L, _L, _L, _L, _P],
    "mbvae_reverse_sample": [_P] * 18 + [_L, _L, _L, _L, _P],
}
_RET64 = {"mbvae_edge_attn_bwd_workspace", "mbvae_ln_residual_bwd_workspace", "mbvae_masked_ce_workspace", "mbvae_kl_workspace"}

_STATUS = {-1: "invalid argument", -2: "unsupported shape", -3: "kernel launch failure"}
_lib = None


def load():
    """Load the shared library once; raises if it has not been built (python mb-vae-dti_amd/build.py)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} not found: build the HIP extension first (python mb-vae-dti_amd/build.py). "
                           "There is no fallback path.")
    # torch bundles its own HIP runtime (libamdhip64); it must be in the process BEFORE this library is loaded so
    # that both share one runtime instance (device pointers and streams are only meaningful within one).
    import torch  # noqa: F401
    lib = ctypes.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)            # AttributeError if a declared symbol is not exported
        fn.argtypes = argtypes
        fn.restype = c_int64 if name in _RET64 else c_int
    _lib = lib
    return lib


# When set to a dict, every kernel call is bracketed by HIP events on the current torch stream:
# PROFILE[(name, tag)] = [(start_event, end_event), ...]   (used by bench.py for per-kernel timings)
PROFILE = None


def call(name, *args, tag=None, allow_unsupported=False):
    """Runs one entry point and raises on failure.  With allow_unsupported, an 'unsupported shape' status is not an
    error: the function returns False and the caller takes its other HIP route (never a CPU one)."""
    fn = getattr(load(), name)
    if PROFILE is None:
        st = fn(*args)
    else:
        import torch
        start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        start.record()
        st = fn(*args)
        end.record()
        if st == 0:
            PROFILE.setdefault((name, tag), []).append((start, end))
    if st == -2 and allow_unsupported:
        return False
    if st != 0:
        raise RuntimeError(f"{name} failed: {_STATUS.get(st, st)}")
    return True


class option:
    """Context manager: `with option("NO_EDGE_FUSED", 1): ...` flips a process-wide tuning switch of the library
    (mbvae_set_option) and restores the previous value on exit."""

    def __init__(self, name, value):
        self.name, self.value = name.encode(), int(value)

    def __enter__(self):
        lib = load()
        self.old = lib.mbvae_get_option(self.name)
        if lib.mbvae_set_option(self.name, self.value) != 0:
            raise ValueError(f"unknown option {self.name!r}")
        return self

    def __exit__(self, *exc):
        load().mbvae_set_option(self.name, self.old)
        return False


def query(name, *args):
    return int(getattr(load(), name)(*args))
