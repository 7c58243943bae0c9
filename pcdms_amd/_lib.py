"""ctypes binding of libpcdm.so (include/pcdm.h).

The product path has exactly one implementation of every op: the hipcc-built gfx950 library.  If
it is missing this module raises -- there is no eager / CPU fallback.  (``use_library`` exists so
tests can inject the lane-emulator build of the *same sources*; the product never calls it.)
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path
from typing import Optional

# torch FIRST: libpcdm.so needs libamdhip64.so.7; PyTorch-ROCm bundles its own copy, and the process must end up
# with ONE HIP runtime (torch's), i.e. torch's must already be loaded when libpcdm.so is dlopen'ed.  Loading
# libpcdm.so first pulls /opt/rocm's runtime in beside torch's and every launch then fails with hipErrorNoDevice.
import torch  # noqa: F401

_HERE = Path(__file__).resolve().parent
LIB_PATH = Path(os.environ.get("PCDM_LIB") or _HERE / "lib" / "libpcdm.so")   # (env: another build of the SAME library, for same-box A/B runs)

_lib: Optional[C.CDLL] = None
_is_emu = False


class GemmParams(C.Structure):
    """Mirror of ``pcdm_gemm_params`` (include/pcdm.h)."""

    _fields_ = [
        ("struct_size", C.c_uint32),
        ("a", C.c_void_p), ("a2", C.c_void_p), ("lda", C.c_int64), ("lda2", C.c_int64),
        ("c1", C.c_int32), ("conv", C.c_int32),
        ("B", C.c_int32), ("Hi", C.c_int32), ("Wi", C.c_int32), ("Ho", C.c_int32), ("Wo", C.c_int32),
        ("stride", C.c_int32), ("upsample", C.c_int32), ("cin", C.c_int32),
        ("w", C.c_void_p),
        ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32), ("Npad", C.c_int32),
        ("bias", C.c_void_p), ("rowvec", C.c_void_p), ("ldrv", C.c_int64), ("rows_per_batch", C.c_int32),
        ("residual", C.c_void_p), ("ldr", C.c_int64), ("res_mod", C.c_int32),
        ("epilogue", C.c_int32), ("vt_col0", C.c_int32),
        ("out", C.c_void_p), ("ldo", C.c_int64), ("out2", C.c_void_p), ("ldo2", C.c_int64),
        ("split_k", C.c_int32), ("ws", C.c_void_p), ("ws_floats", C.c_int64),
        ("ldw", C.c_int64), ("no_pad_lo", C.c_int32), ("tile", C.c_int32), ("act", C.c_int32),
        ("zero_rows", C.c_int32),
        ("ln_wsum", C.c_void_p), ("ln_eps", C.c_float), ("defer_reduce", C.c_int32),
        ("rowvec_step", C.c_void_p), ("rowvec_step_stride", C.c_int64), ("dup_rows", C.c_int32),
        ("ln_row_stats", C.c_void_p), ("row_stats_out", C.c_void_p), ("rowvec_step_count", C.c_int32), ("step_error", C.c_void_p),
        ("a3", C.c_void_p), ("lda3", C.c_int64), ("tap_lut", C.c_uint64), ("tap_group_n", C.c_int32),
    ]

    def __init__(self, *args, **kw):
        super().__init__(*args, **kw)
        self.struct_size = C.sizeof(GemmParams)   # (ABI 4: pcdm_gemm refuses a struct of another size)


class GnSplitKSrc(C.Structure):
    """Mirror of ``pcdm_gn_splitk_src`` (include/pcdm.h)."""

    _fields_ = [("part", C.c_void_p), ("split_k", C.c_int32), ("M", C.c_int32), ("N", C.c_int32), ("Npad", C.c_int32),
                ("bias", C.c_void_p), ("rowvec", C.c_void_p), ("ldrv", C.c_int64), ("rowvec_step", C.c_void_p), ("rowvec_step_stride", C.c_int64),
                ("residual", C.c_void_p), ("ldr", C.c_int64), ("pre_out", C.c_void_p), ("store_pre", C.c_int32),
                ("rowvec_step_count", C.c_int32), ("step_error", C.c_void_p)]


class UNetConfig(C.Structure):
    """Mirror of ``pcdm_unet_config`` (include/pcdm.h)."""

    _fields_ = [("out_channels", C.c_int32), ("n_levels", C.c_int32), ("block_out_channels", C.c_int32 * 8), ("heads", C.c_int32 * 8),
                ("cross_attn", C.c_int32 * 8), ("layers_per_block", C.c_int32), ("cross_attention_dim", C.c_int32),
                ("norm_groups", C.c_int32), ("norm_eps", C.c_float), ("class_embed", C.c_int32), ("flip_sin_to_cos", C.c_int32),
                ("freq_shift", C.c_float)]


_P, _I, _L, _F = C.c_void_p, C.c_int, C.c_int64, C.c_float
_SIGS = {
    "pcdm_version": ([], C.c_int),
    "pcdm_is_emulator": ([], C.c_int),
    "pcdm_groupnorm_ws_floats": ([_I, _I], _L),
    "pcdm_groupnorm_cluster_timeouts": ([_P, C.POINTER(C.c_uint), _P], C.c_int),
    "pcdm_groupnorm": ([_P, _I, _P, _I, _I, _I, _I, _F, _P, _P, _I, _P, _P, _P], C.c_int),
    "pcdm_groupnorm_splitk": ([C.POINTER(GnSplitKSrc), _P, _I, _I, _I, _I, _F, _P, _P, _I, _P, _P, _P], C.c_int),
    "pcdm_layernorm": ([_P, _P, _I, _I, _F, _P, _P, _P], C.c_int),
    "pcdm_gemm": ([C.POINTER(GemmParams), _P], C.c_int),
    "pcdm_flash_attn": ([_P, _L, _P, _L, _P, _L, _P, _L, _I, _I, _I, _I, _F, _P], C.c_int),
    "pcdm_flash_attn_thr": ([_P, _L, _P, _L, _P, _L, _P, _L, _I, _I, _I, _I, _F, _F, _P], C.c_int),
    "pcdm_quantize_fp8": ([_P, _P, _L, _I, _I, _L, _L, _F, _P], C.c_int),
    "pcdm_flash_attn_fp8": ([_P, _L, _P, _L, _P, _L, _P, _L, _I, _I, _I, _I, _F, _F, _F, _F, _P], C.c_int),
    "pcdm_timestep_embedding": ([_P, _P, _P, _I, _I, _I, _F, _P], C.c_int),
    "pcdm_timestep_embedding_rows": ([_P, _I, _P, _I, _I, _F, _P], C.c_int),
    "pcdm_time_class_combine": ([_P, _P, _P, _I, _I, _I, _P], C.c_int),
    "pcdm_small_linear": ([_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P], C.c_int),
    "pcdm_assemble_input": ([_P, _I, _I, _P, _I, _P, _I, _P, _I, _I, _I, _P], C.c_int),
    "pcdm_nchw_f32_to_nhwc_bf16": ([_P, _P, _I, _I, _I, _I, _P], C.c_int),
    "pcdm_nhwc_bf16_to_nchw_f32": ([_P, _P, _I, _I, _I, _P], C.c_int),
    "pcdm_f32_to_bf16": ([_P, _P, _L, _P], C.c_int),
    "pcdm_cfg_step": ([_P, _I, _F, _P, _P, _P, _P, _P, _P, _L, _P], C.c_int),
    "pcdm_unipc_step": ([_P, _I, _F, _P, _P, _P, _P, _P, _P, _L, _P], C.c_int),
    "pcdm_unclip_step": ([_P, _I, _F, _P, _P, _P, C.POINTER(_F), _L, _P], C.c_int),
    "pcdm_unclip_step_dev": ([_P, _I, _F, _P, _P, _P, _P, _L, _P], C.c_int),
    "pcdm_lincomb": ([_P, _I, C.POINTER(_P), C.POINTER(_F), _L, _P], C.c_int),
    "pcdm_rescale_noise_cfg": ([_P, _P, _P, _I, _L, _F, _P], C.c_int),
    "pcdm_softmax_rows": ([_P, _P, _I, _I, _L, _L, _F, _P], C.c_int),
    "pcdm_gaussian_sample": ([_P, _P, _P, _I, _I, _I, _F, _P], C.c_int),
    "pcdm_image_to_uint8": ([_P, _P, _I, _I, _I, _P], C.c_int),
    "pcdm_advance_step": ([_P, _P], C.c_int),
    "pcdm_pixel_shuffle2": ([_P, _P, _I, _I, _I, _I, _P], C.c_int),
    "pcdm_unet_create": ([C.POINTER(UNetConfig)], _P),
    "pcdm_unet_destroy": ([_P], None),
    "pcdm_unet_last_error": ([_P], C.c_char_p),
    "pcdm_unet_set_weight": ([_P, C.c_char_p, _P, _P, _P, _I, _I, _I, _I], C.c_int),
    "pcdm_unet_set_vector": ([_P, C.c_char_p, _P, _I], C.c_int),
    "pcdm_unet_set_tile": ([_P] + [_I] * 13, C.c_int),
    "pcdm_unet_set_attention_fp8": ([_P, _I], C.c_int),
    "pcdm_unet_get_tile": ([_P] + [_I] * 11 + [C.POINTER(C.c_int), C.POINTER(C.c_int)], C.c_int),
    "pcdm_unet_workspace_bytes": ([_P, _I, _I, _I, _I], _L),
    "pcdm_unet_workspace_init": ([_P, _I, _I, _I, _I, _P, _P], C.c_int),
    "pcdm_unet_prepare_conditioning": ([_P, _I, _I, _I, _I, _P, _P, _P, _I, _I, _P, _P], C.c_int),
    "pcdm_unet_time_table_bytes": ([_P, _I, _I], _L),
    "pcdm_unet_prepare_timesteps": ([_P, _P, _I, _P, _P, _P], C.c_int),
    "pcdm_unet_set_shared_cfg_input": ([_P, _P, _I], C.c_int),
    "pcdm_unet_forward": ([_P, _P, _P, _P, _I, _I, _I, _I, _I, _P, _P, _P], C.c_int),
    "pcdm_unet_step_overflow": ([_P, _P, C.POINTER(C.c_int), _P], C.c_int),
    "pcdm_pack_linear": ([_P, _P, _I, _I, _I, _P, _P], C.c_int),
    "pcdm_pack_conv3x3": ([_P, _P, _I, _I, _I, _P, _P, _P, _P], C.c_int),
    "pcdm_pack_geglu": ([_P, _P, _I, _I, _P, _P], C.c_int),
}
EXPORTS = tuple(_SIGS)


def _bind(lib: C.CDLL) -> C.CDLL:
    for name, (args, res) in _SIGS.items():
        fn = getattr(lib, name)  # AttributeError if the library does not export it
        fn.argtypes = args
        fn.restype = res
    return lib


def use_library(lib: C.CDLL) -> None:
    """Install an already-loaded library (tests: the emulator build of the same sources)."""
    global _lib, _is_emu
    _lib = _bind(lib)
    _is_emu = bool(lib.pcdm_is_emulator())


def load(path: Optional[Path] = None) -> C.CDLL:
    global _lib, _is_emu
    p = Path(path) if path else LIB_PATH
    if not p.exists():
        raise RuntimeError(
            f"{p} not found: the HIP kernel library is not built.  Run `python -m pcdms_amd.build` "
            "(needs hipcc); pcdms_amd has no fallback implementation.")
    _lib = _bind(C.CDLL(str(p)))
    _is_emu = bool(_lib.pcdm_is_emulator())
    return _lib


def lib() -> C.CDLL:
    return _lib if _lib is not None else load()


def is_emulator() -> bool:
    lib()
    return _is_emu
