"""ctypes binding of libuse_hip.so (C ABI: include/use_hip.h).

There is no CPU fallback: if the shared library is missing or fails to load, importing the product path
raises -- build it with ``python -c "import __graft_entry__ as g; g.build()"`` or ``make -C
universal_speech_enhancement_amd/csrc``.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("USE_HIP_LIB") or os.path.join(_HERE, "libuse_hip.so")   # USE_HIP_LIB: same-box A/B of two builds

PREC = {"fp32": 0, "bf16": 1, "fp16": 2}
PREDICTORS = {"reverse_diffusion": 0, "euler_maruyama": 1, "none": 2}
CORRECTORS = {"none": 0, "langevin": 1, "ald": 2}


class UseConfig(C.Structure):
    _fields_ = [("nf", C.c_int), ("n_levels", C.c_int), ("ch_mult", C.c_int * 8), ("num_res_blocks", C.c_int),
                ("n_freq", C.c_int), ("precision", C.c_int), ("theta", C.c_float), ("sigma_min", C.c_float),
                ("sigma_max", C.c_float), ("input_channels", C.c_int), ("unconditional", C.c_int),
                ("no_sigma_scale", C.c_int)]


class UseSamplerConfig(C.Structure):
    _fields_ = [("N", C.c_int), ("predictor", C.c_int), ("corrector", C.c_int), ("corrector_steps", C.c_int),
                ("snr", C.c_float), ("t_eps", C.c_float), ("use_graph", C.c_int)]


class UseConvCase(C.Structure):
    _fields_ = [(k, C.c_int) for k in ("B", "H", "W", "C0", "C1", "Cout", "XC0", "XC1", "act", "gn", "temb", "res", "stats",
                                       "dtype", "variant", "iters")]


class UseConvOp(C.Structure):
    _fields_ = ([(k, C.c_int) for k in ("B", "H", "W", "C0", "C1", "Cout", "XC0", "XC1", "ntaps", "act", "dtype", "out_dtype", "variant")] +
                [(k, C.c_void_p) for k in ("src0", "src1", "coef", "w", "bias", "temb", "x0", "x1", "w2", "res")] +
                [("out_scale", C.c_float), ("out", C.c_void_p), ("stats", C.c_void_p)])


class UseHipError(RuntimeError):
    pass


_lib = None

# every symbol declared in include/use_hip.h: (restype, argtypes)
_vp, _i, _i64, _u64, _f = C.c_void_p, C.c_int, C.c_int64, C.c_uint64, C.c_float
SYMBOLS = {
    "use_set_option": (C.c_int, [C.c_char_p, C.c_longlong]),
    "use_last_error": (C.c_char_p, []),
    "use_version": (C.c_char_p, []),
    "use_create": (_i, [C.POINTER(UseConfig), _i, C.POINTER(_vp)]),
    "use_destroy": (_i, [_vp]),
    "use_set_weight": (_i, [_vp, C.c_char_p, _vp, C.POINTER(_i64), _i]),
    "use_commit_weights": (_i, [_vp]),
    "use_alloc_weight_blob": (_i, [_vp]),
    "use_weight_blob": (_i, [_vp, C.POINTER(_vp), C.POINTER(C.c_size_t)]),
    "use_save_weight_blob": (C.c_int, [C.c_void_p, C.c_char_p]),
    "use_load_weight_blob": (C.c_int, [C.c_void_p, C.c_char_p]),
    "use_num_expected_weights": (_i, [_vp]),
    "use_expected_weight": (_i, [_vp, _i, C.POINTER(C.c_char_p), C.POINTER(_i64), C.POINTER(_i)]),
    "use_plan": (_i, [_vp, _i, _i]),
    "use_workspace_bytes": (_i, [_vp, C.POINTER(C.c_size_t)]),
    "use_score": (_i, [_vp, _vp, _vp, _vp, _vp, _vp]),
    "use_forward": (_i, [_vp, _vp, _vp, _vp, _vp, _vp]),
    "use_score2": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "use_sample_cond2": (_i, [_vp, _vp, _vp, _vp, _vp, _u64, _vp, _vp]),
    "use_set_sampler": (_i, [_vp, C.POINTER(UseSamplerConfig)]),
    "use_num_noise_draws": (_i, [_vp]),
    "use_get_timesteps": (_i, [_vp, C.POINTER(_f), _i]),
    "use_sample": (_i, [_vp, _vp, _vp, _u64, _vp, _vp]),
    "use_sample_cond": (_i, [_vp, _vp, _vp, _vp, _u64, _vp, _vp]),
    "use_spec_fwd": (_i, [_vp, _vp, _i, _i, _i, _i, C.c_float, C.c_float, _vp]),
    "use_spec_back": (_i, [_vp, _vp, _i, _i, _i, _i, C.c_float, C.c_float, _vp]),
    "use_stft_fwd": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _i, C.c_float, C.c_float, _vp]),
    "use_istft_back": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _i, C.c_float, C.c_float, _vp]),
    "use_sde_prior": (_i, [_vp, _vp, _vp, _u64, _vp, _i64, _vp]),
    "use_fill_noise": (_i, [_vp, _u64, _i, _vp, _i64, _vp]),
    "use_sde_predictor": (_i, [_vp, _i, _f, _i, _vp, _vp, _vp, _vp, _u64, _vp, _vp, _i64, _vp]),
    "use_sde_corrector": (_i, [_vp, _i, _f, _f, _i, _vp, _vp, _vp, _u64, _vp, _vp, _i64, _vp]),
    "use_debug_tensor": (_i, [_vp, C.c_char_p, C.POINTER(_vp), C.POINTER(_i), C.POINTER(_i)]),
    "use_flops_per_score": (C.c_double, [_vp]),
    "use_get_stat": (_i, [_vp, C.c_char_p, C.POINTER(C.c_longlong)]),
    "use_profile_aux": (_i, [_vp, _i, C.c_char_p, _i, C.POINTER(_i), C.POINTER(_i), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "use_profile_aux_flops": (_i, [_vp, _i, C.POINTER(C.c_double)]),
    "use_profile_score": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(_i), C.POINTER(C.c_double)]),
    "use_timesteps": (_i, [_i, _f, C.POINTER(_f)]),
    "use_conv_bench": (_i, [C.POINTER(UseConvCase), _vp, _vp, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "use_op_conv": (_i, [C.POINTER(UseConvOp), _vp]),
    "use_op_conv_dev_workspace": (C.c_size_t, [C.POINTER(UseConvOp)]),
    "use_op_conv_dev": (_i, [C.POINTER(UseConvOp), _i, _vp, C.c_size_t, _vp]),
    "use_op_fir": (_i, [_vp, _i, _vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "use_op_attention": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "use_op_gn_finalize": (_i, [_vp, _i, _vp, _i, _vp, _vp, _i, _i, _f, _vp, _i, _vp]),
    "use_op_wgrad_workspace": (C.c_size_t, [_i, _i, _i, _i, _i, _i, _i]),
    "use_op_wgrad": (_i, [_vp, _vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _i, _f, _vp, C.c_size_t, _vp]),
    "use_op_gn_workspace": (C.c_size_t, [_i, _i, _i]),
    "use_op_gn_act_bwd": (_i, [_vp, _vp, _i, _vp, _vp, _i, _f, _i, _vp, _f, _i, _i, _i, _vp, _i, _vp, _vp, _vp, _vp]),
    "use_op_gn_act_fwd": (_i, [_vp, _i, _vp, _vp, _i, _f, _i, _i, _i, _i, _vp, _vp, _vp]),
    "use_op_colsum": (_i, [_vp, _i, _i, _i, _i, _f, _vp, _vp, _vp]),
    "use_op_attention_bwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "use_op_dense_bwd": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "use_wav_read": (_i, [C.c_char_p, C.POINTER(C.POINTER(C.c_double)), C.POINTER(_i64), C.POINTER(_i), C.POINTER(_i)]),
    "use_wav_write": (_i, [C.c_char_p, _vp, _i64, _i, _i, _i]),
    "use_resample_fft": (_i, [_vp, _i64, _i64, _vp]),
    "use_load_utterance": (_i, [C.c_char_p, _i, _i, C.POINTER(C.POINTER(C.c_float)), C.POINTER(_i64), C.POINTER(_i)]),
    "use_free": (None, [_vp]),
}


def lib() -> C.CDLL:
    """Load (once) and return the shared library with typed prototypes; raises if it is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise UseHipError(
                f"{LIB_PATH} not found: the HIP extension is not built (run __graft_entry__.build()). "
                "There is no CPU fallback for the sampling path.")
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            if os.environ.get("USE_HIP_LIB") and not hasattr(L, name):
                continue               # an older build under A/B test may lack the newest entry points
            fn = getattr(L, name)  # AttributeError if the .so does not export a declared symbol
            fn.restype, fn.argtypes = res, args
        # USE_HIP_OPTS="name=value,...": use_set_option calls at load time (A/B runs of the test suite and the bench with an option flipped);
        # applied BEFORE the library is published: a bad entry fails every lib() call, not only the first
        for kv in filter(None, os.environ.get("USE_HIP_OPTS", "").split(",")):
            k, sep, v = kv.partition("=")
            if not sep or not k.strip():
                raise UseHipError(f"USE_HIP_OPTS: entry {kv!r} is not name=value")
            try:
                iv = int(v)
            except ValueError:
                raise UseHipError(f"USE_HIP_OPTS: {kv!r}: the value must be an integer") from None
            if L.use_set_option(k.strip().encode(), iv) < 0:
                raise UseHipError(f"USE_HIP_OPTS: {kv}: " + L.use_last_error().decode(errors="replace"))
        _lib = L
    return _lib


def check(rc: int, what: str = "") -> int:
    if rc < 0:
        msg = lib().use_last_error().decode(errors="replace")
        raise UseHipError(f"{what or 'libuse_hip'} failed ({rc}): {msg}")
    return rc
