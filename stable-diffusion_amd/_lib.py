"""ctypes binding of csrc/libsdamd.so (declarations: include/sdamd.h).

There is deliberately no fallback: if the library is missing, every op raises.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# SDAMD_LIB: another build of the same ABI (A/B experiments); never a fallback -- it must exist too
LIB_PATH = os.environ.get("SDAMD_LIB") or os.path.join(_HERE, "csrc", "libsdamd.so")

_vp, _i, _f, _ll, _sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_longlong, ctypes.c_size_t

# name -> (restype, argtypes); mirrors include/sdamd.h one to one
SIGNATURES = {
    "sdamd_last_error": (ctypes.c_char_p, []),
    "sdamd_abi_version": (_i, []),
    "sdamd_device_count": (_i, []),
    "sdamd_cfg_ddpm_step": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _f, _i, _i, _i, _vp]),
    "sdamd_counter_add": (_i, [_vp, _i, _vp]),
    "sdamd_add_noise": (_i, [_vp, _vp, _vp, _f, _f, _ll, _vp]),
    "sdamd_gather_row": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp]),
    "sdamd_groupnorm_ws_bytes": (_sz, [_i]),
    "sdamd_groupnorm_nhwc": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _f, _i, _i, _i, _vp]),
    "sdamd_noalloc_init": (_i, []),
    "sdamd_groupnorm_noalloc_max_batch": (_i, []),
    "sdamd_groupnorm_nhwc_noalloc": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _f, _i, _i, _i, _vp]),
    "sdamd_groupnorm_nhwc_pre": (_i, [_vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _f, _i, _i, _i, _vp]),
    "sdamd_groupnorm_nhwc_cat": (_i, [_vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _i, _i, _f, _i, _i, _i, _vp]),
    "sdamd_layernorm": (_i, [_vp, _vp, _vp, _vp, _i, _i, _f, _i, _i, _vp]),
    "sdamd_geglu": (_i, [_vp, _vp, _ll, _i, _i, _vp]),
    "sdamd_clip_embed": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "sdamd_concat_nhwc": (_i, [_vp, _vp, _vp, _ll, _i, _i, _i, _vp]),
    "sdamd_add": (_i, [_vp, _vp, _vp, _ll, _i, _vp]),
    "sdamd_nchw_to_nhwc": (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    "sdamd_nhwc_to_nchw": (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    "sdamd_cast": (_i, [_vp, _vp, _ll, _i, _i, _vp]),
    "sdamd_split_hilo": (_i, [_vp, _vp, _ll, _i, _vp]),
    "sdamd_split_hilo_cat": (_i, [_vp, _vp, _vp, _ll, _i, _i, _vp]),
    "sdamd_silu": (_i, [_vp, _vp, _ll, _vp]),
    "sdamd_image_to_uint8": (_i, [_vp, _vp, _i, _i, _vp]),
    "sdamd_uint8_to_image": (_i, [_vp, _vp, _i, _i, _vp]),
    "sdamd_vae_post_quant": (_i, [_vp, _vp, _vp, _vp, _i, _i, _vp]),
    "sdamd_vae_sample": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _vp]),
    "sdamd_set_plan_samples": (_i, [_i]),
    "sdamd_gemm_ln": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _f, _i, _i, _i, _i, _vp]),
    "sdamd_gemm": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _ll, _ll, _ll, _ll, _i, _i, _i, _vp, _sz, _vp, _vp, _vp, _vp]),
    "sdamd_split3_weights": (_i, [_vp, _vp, _ll, _i, _i, _vp]),
    "sdamd_gemm_tile": (_i, [_i, _i, _i, _i, _i, _sz, _i]),
    "sdamd_conv3x3_halo": (_i, [_i] * 9 + [_sz]),
    "sdamd_set_gemm_impl": (_i, [_i]),
    "sdamd_set_f32_mode": (_i, [_i]),
    "sdamd_gemm_supports_act": (_i, [_i, _i, _i]),
    "sdamd_conv3x3": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _sz, _vp, _vp, _vp, _vp]),
    "sdamd_conv3x3_cin_small": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "sdamd_conv3x3_cout_small": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "sdamd_attention": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _ll, _i, _ll, _i, _ll, _i, _ll, _i, _i, _i, _vp]),
}

_lib = None


class SdamdError(RuntimeError):
    pass


def lib():
    """Load (once) and return the ctypes handle; raises if the HIP library was not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise SdamdError(
                f"{LIB_PATH} not found: build it with `make -C stable-diffusion_amd/csrc` "
                "(or __graft_entry__.build()). There is no CPU fallback.")
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)   # AttributeError if the .so does not export a declared symbol
            fn.restype = res
            fn.argtypes = args
        if os.environ.get("SDAMD_F32_MODE", "") == "exact":   # exact f32 MFMA instead of the bf16x3 products (include/sdamd.h)
            handle.sdamd_set_f32_mode(0)
        _lib = handle
    return _lib


def check(status):
    if status != 0:
        raise SdamdError(lib().sdamd_last_error().decode("utf-8", "replace"))
