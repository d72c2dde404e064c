"""ctypes loader for libvadc_amd.so (the C-ABI of include/vadc_amd.h).

The library is built in-tree by `make -C vadc_amd/csrc` (or __graft_entry__.build()).  There is no Python or
CPU fallback: if the shared object is missing this module raises at import of the symbols.
"""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libvadc_amd.so")

# every symbol include/vadc_amd.h declares (tests/test_abi.py checks header <-> library <-> this list)
ABI_VERSION = 6          # VADC_AMD_ABI_VERSION of the include/vadc_amd.h this module's Caps structure mirrors (tests/test_abi.py compares the two)
SYMBOLS = [
    "vadc_amd_abi_version", "vadc_amd_create", "vadc_amd_destroy", "vadc_amd_last_error", "vadc_amd_get_caps", "vadc_amd_get_caps_sized",
    "vadc_amd_run_f32", "vadc_amd_run_s16", "vadc_amd_run_device_f32", "vadc_amd_run_device_s16",
    "vadc_amd_run_s16_async", "vadc_amd_run_f32_async", "vadc_amd_wait_async",
    "vadc_amd_synchronize", "vadc_amd_join", "vadc_amd_speech_probabilities", "vadc_amd_reset_streams", "vadc_amd_get_state", "vadc_amd_set_state",
    "vadc_amd_get_context", "vadc_amd_set_context",
    "vadc_amd_debug_stage_from_samples", "vadc_amd_debug_stage_from_stage", "vadc_amd_debug_lstm_decoder", "vadc_amd_debug_layer1_block", "vadc_amd_debug_decoder", "vadc_amd_unpin",
    "vadc_amd_set_option", "vadc_amd_get_option", "vadc_amd_set_profiling", "vadc_amd_get_kernel_time", "vadc_amd_reset_kernel_times",
    "vadc_amd_kernel_name",
]


class Caps(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "batch_size_restriction", "is_silero_v5", "input_size_min", "input_size_max", "output_dims",
        "output_stride", "silero_probability_out_index", "lstm_hidden_size", "max_streams",
        "max_chunks_per_call", "device", "precision", "model_kind", "lstm_steps_per_chunk", "window_samples", "sample_rate", "context_size", "cu_partition_ok", "input_size_step")]


_lib = None


def _preload_hip_runtime() -> None:
    """One HIP runtime per process.  libvadc_amd.so needs `libamdhip64.so.7` (system ROCm by default).  A
    PyTorch-ROCm wheel bundles its own copy and asks for it by the unversioned file name, so if OUR library
    pulled in the system runtime first, torch would later load a second runtime and find no GPU.  When torch
    is installed, load its bundled runtime first (by SONAME both resolve to it); C hosts never get here and
    use the system runtime through DT_NEEDED.  Set VADC_AMD_HIP_RUNTIME=system to skip."""
    if os.environ.get("VADC_AMD_HIP_RUNTIME", "") == "system":
        return
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.submodule_search_locations:
            return
        cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
        if os.path.exists(cand):
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
    except Exception:
        pass


def load() -> C.CDLL:
    global _lib, LIB_PATH
    if _lib is not None:
        return _lib
    # experiments only (tools/abl_build.sh): a variant of the library built beside the product, never in its place
    if os.environ.get("VADC_AMD_LIB"):
        LIB_PATH = os.environ["VADC_AMD_LIB"]
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build the HIP extension first (python -c 'import __graft_entry__ as g; g.build()' "
            "or make -C vadc_amd/csrc). vadc_amd has no CPU fallback.")
    _preload_hip_runtime()
    L = C.CDLL(LIB_PATH)
    vp, i32, f32p = C.c_void_p, C.c_int, C.POINTER(C.c_float)
    L.vadc_amd_create.argtypes = [C.c_char_p, C.c_size_t, i32, i32, i32, i32, C.POINTER(vp)]
    L.vadc_amd_destroy.argtypes = [vp]
    L.vadc_amd_destroy.restype = None
    L.vadc_amd_last_error.restype = C.c_char_p
    L.vadc_amd_get_caps.argtypes = [vp, C.POINTER(Caps)]
    L.vadc_amd_run_f32.argtypes = [vp, vp, i32, i32, vp]
    L.vadc_amd_run_s16.argtypes = [vp, vp, i32, i32, vp]
    L.vadc_amd_run_device_f32.argtypes = [vp, vp, i32, i32, vp, vp]
    L.vadc_amd_run_device_s16.argtypes = [vp, vp, i32, i32, vp, vp]
    L.vadc_amd_run_s16_async.argtypes = [vp, vp, i32, i32, vp]
    L.vadc_amd_run_f32_async.argtypes = [vp, vp, i32, i32, vp]
    L.vadc_amd_wait_async.argtypes = [vp]
    L.vadc_amd_get_context.argtypes = [vp, i32, vp]
    L.vadc_amd_set_context.argtypes = [vp, i32, vp]
    L.vadc_amd_synchronize.argtypes = [vp]
    L.vadc_amd_join.argtypes = [vp, vp]
    L.vadc_amd_speech_probabilities.argtypes = [vp, vp, i32, i32, vp, vp]
    L.vadc_amd_reset_streams.argtypes = [vp, vp, i32]
    L.vadc_amd_get_state.argtypes = [vp, i32, vp, vp]
    L.vadc_amd_set_state.argtypes = [vp, i32, vp, vp]
    L.vadc_amd_debug_stage_from_samples.argtypes = [vp, vp, i32, i32, vp]
    L.vadc_amd_debug_stage_from_stage.argtypes = [vp, vp, i32, i32, i32, vp]
    L.vadc_amd_debug_lstm_decoder.argtypes = [vp, vp, i32, i32, vp]
    L.vadc_amd_debug_layer1_block.argtypes = [vp, i32, vp, i32, vp]
    L.vadc_amd_debug_decoder.argtypes = [vp, vp, i32, vp]
    L.vadc_amd_unpin.argtypes = [vp, vp]
    L.vadc_amd_set_option.argtypes = [vp, C.c_char_p, i32]
    L.vadc_amd_get_option.argtypes = [vp, C.c_char_p, C.POINTER(C.c_int32)]
    L.vadc_amd_set_profiling.argtypes = [vp, i32]
    L.vadc_amd_get_kernel_time.argtypes = [vp, i32, C.POINTER(i32), C.POINTER(C.c_double)]
    L.vadc_amd_reset_kernel_times.argtypes = [vp]
    L.vadc_amd_kernel_name.argtypes = [i32]
    L.vadc_amd_kernel_name.restype = C.c_char_p
    for name in SYMBOLS:
        getattr(L, name)          # AttributeError here == header/library mismatch
    L.vadc_amd_abi_version.restype = C.c_int
    if L.vadc_amd_abi_version() != ABI_VERSION:      # before any struct crosses the boundary: get_caps writes the LIBRARY's sizeof(vadc_amd_caps)
        raise RuntimeError(f"{LIB_PATH} speaks revision {L.vadc_amd_abi_version()} of include/vadc_amd.h, vadc_amd/_lib.py revision {ABI_VERSION}: rebuild (make -C vadc_amd/csrc)")
    _lib = L
    return L
