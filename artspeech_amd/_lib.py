"""ctypes binding of libartspeech_hip.so (the C ABI declared in include/artspeech_hip.h).

There is NO CPU fallback: if the library is missing every op raises.  torch is imported first so that
the HIP runtime already loaded by torch (its bundled libamdhip64) is the one the library binds to.
"""
import ctypes as C
import os

import torch  # noqa: F401  (must be loaded before the library: single HIP runtime per process)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libartspeech_hip.so")
# tools/ only: the -DAS_DIAG flavour (ablation switches, legacy kernels; `python -m artspeech_amd.build --diag`).  The
# product library has none of them compiled in, so AS_* environment variables cannot change what it computes.
if os.environ.get("ARTSPEECH_DIAG_LIB") == "1":
    LIB_PATH = os.path.join(_HERE, "libartspeech_hip_diag.so")
elif os.environ.get("ARTSPEECH_DIAG_LIB"):   # a diagnostic flavour built with extra defines (build.py --suffix=...)
    LIB_PATH = os.path.abspath(os.environ["ARTSPEECH_DIAG_LIB"])

c_f32p = C.c_void_p  # device pointers travel as integers (tensor.data_ptr())


class Dims(C.Structure):
    _fields_ = [("vocab", C.c_int32), ("n_art", C.c_int32), ("embed", C.c_int32), ("hidden", C.c_int32),
                ("n_samp", C.c_int32), ("simple", C.c_int32)]


class Layout(C.Structure):
    _fields_ = [("embedding", C.c_int64),
                ("w_ih", C.c_int64 * 2), ("b_ih", C.c_int64 * 2), ("w_hh", C.c_int64 * 2), ("b_hh", C.c_int64 * 2),
                ("lin_w", C.c_int64), ("lin_b", C.c_int64),
                ("ln1_g", C.c_int64), ("ln1_b", C.c_int64), ("w1", C.c_int64), ("b1", C.c_int64),
                ("ln2_g", C.c_int64), ("ln2_b", C.c_int64), ("w2", C.c_int64), ("b2", C.c_int64),
                ("ln3_g", C.c_int64), ("ln3_b", C.c_int64), ("w3", C.c_int64), ("b3", C.c_int64),
                ("total", C.c_int64)]


class Opts(C.Structure):
    _fields_ = [("gru_dropout", C.c_float), ("dropout_seed", C.c_uint64), ("dout_presigmoid", C.c_int32),
                ("defer_dw2", C.c_int32), ("fold_wait_event", C.c_void_p),
                ("loss_targets", C.c_void_p), ("loss_tgt_T", C.c_int64), ("loss_scale", C.c_float), ("loss_out", C.c_void_p),
                ("loss_dout", C.c_void_p)]


class Gemm(C.Structure):
    _fields_ = [("A", C.c_void_p), ("B", C.c_void_p), ("C", C.c_void_p), ("bias", C.c_void_p),
                ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32),
                ("a_i", C.c_int64), ("a_k", C.c_int64), ("b_j", C.c_int64), ("b_k", C.c_int64), ("ldc", C.c_int64),
                ("batch", C.c_int32), ("a_batch", C.c_int64), ("b_batch", C.c_int64), ("c_batch", C.c_int64),
                ("bias_batch", C.c_int64), ("act", C.c_int32), ("accumulate", C.c_int32),
                ("b_kshift", C.c_int32), ("b_kT", C.c_int32),
                ("splitk_ws", C.c_void_p), ("splitk_ws_floats", C.c_int64), ("colsum", C.c_void_p),
                ("colsum_batch", C.c_int64), ("a_off", C.c_void_p), ("b_off", C.c_void_p), ("c_off", C.c_void_p),
                ("bias_off", C.c_void_p), ("precision", C.c_int32), ("b_kshift_batch", C.c_int32), ("cu_budget", C.c_int32),
                ("res", C.c_void_p), ("res_ld", C.c_int64), ("res_batch", C.c_int64), ("res_off", C.c_void_p),
                ("mask_bits", C.c_void_p), ("mask_batch", C.c_int64), ("relu_bits", C.c_void_p), ("relu_bits_batch", C.c_int64),
                ("k_seg", C.c_int32), ("a_seg_off", C.c_void_p), ("b_seg_off", C.c_void_p), ("k_tri", C.c_int32)]


_P, _I32, _I64, _F, _D = C.c_void_p, C.c_int32, C.c_int64, C.c_float, C.c_double
_DIMS, _LAY = C.POINTER(Dims), C.POINTER(Layout)

# name -> (restype, argtypes); every symbol include/artspeech_hip.h declares
PROTOTYPES = {
    "as_version": (C.c_char_p, []),
    "as_arch": (C.c_char_p, []),
    "as_last_error": (C.c_char_p, []),
    "as_artspeech_layout": (_I32, [_DIMS, _LAY]),
    "as_artspeech_workspace_floats": (_I64, [_DIMS, _I32, _I32]),
    "as_artspeech_fwd": (_I32, [_DIMS, _P, _P, _I64, _P, _I32, _I32, _P, _P, _I32, C.POINTER(Opts), _P]),
    "as_artspeech_bwd": (_I32, [_DIMS, _P, _P, _I64, _P, _I32, _I32, _P, _P, _P, _P, C.POINTER(Opts), _P]),
    "as_artspeech_dw2": (_I32, [_DIMS, _P, _I32, _I32, _P, _P, _P]),
    "as_gru_bidir_fwd": (_I32, [_P, _P, _I64, _P, _P, _P, _I32, _I32, _I32, _P, _P, _P]),
    "as_gru_bidir_bwd": (_I32, [_P, _P, _P, _P, _P, _I32, _I32, _I32, _P, _P, _P]),
    "as_gemm_f32": (_I32, [C.POINTER(Gemm), _P]),
    "as_linear_planes_floats": (_I64, [_I32, _I32]),
    "as_linear_fwd": (_I32, [_P, _I64, _P, _I64, _P, _P, _I64, _I32, _I32, _I32, _I32, _P, _P]),
    "as_layernorm_fwd_blockres": (_I32, [_P, _P, _P, _P, _I32, _I64, _I32, _I32, _P]),
    "as_head_workspace_floats": (_I64, [_DIMS, _I64]),
    "as_head_fwd": (_I32, [_DIMS, _LAY, _P, _P, _I64, _P, _P, _I32, _P]),
    "as_head_bwd": (_I32, [_DIMS, _LAY, _P, _P, _P, _I64, _P, _P, _P, _P]),
    "as_euclid_fwd": (_I32, [_P, _P, _I64, _I32, _I32, _P, _P]),
    "as_euclid_bwd": (_I32, [_P, _P, _P, _I64, _I32, _I32, _P, _P]),
    "as_euclid_masked_partials": (_I32, []),
    "as_euclid_masked_fwd_bwd": (_I32, [_P, _P, _I64, _P, _I32, _I32, _I32, _I32, _F, _P, _P, _P, _P]),
    "as_euclid_masked_fwd_bwd_presigmoid": (_I32, [_P, _P, _I64, _P, _I32, _I32, _I32, _I32, _F, _P, _P, _P, _P]),
    "as_p2cp_fwd": (_I32, [_P, _I64, _I64, _I64, _I32, _P, _I64, _I64, _I64, _I32, _I64, _P, _P]),
    "as_p2cp_utterance_mean": (_I32, [_P, _P, _I32, _I32, _I32, _F, _P, _P]),
    "as_pearson_fwd": (_I32, [_P, _I64, _I64, _P, _I64, _I64, _I32, _I32, _I32, _I32, _F, _P, _P, _P]),
    "as_tract_variables_fwd": (_I32, [_P, _I64, _I32, _I32, _P, _I32, _P, _P, _P, _P, _P]),
    "as_area_function_fwd": (_I32, [_P, _P, _I64, _I64, _I64, _I64, _I32, _D, _D, _P, _P, _P]),
    "as_evenly_spaced_fx": (_I32, [_P, _P, _I64, _I32, _I32, _P, _P]),
    "as_adam_step": (_I32, [_P, _P, _P, _P, _I64, _F, _F, _F, _F, _F, _I64, _F, _P]),
    "as_dropout_fwd": (_I32, [_P, _P, _I64, _F, C.c_uint64, _P]),
    "as_layernorm_fwd": (_I32, [_P, _P, _P, _P, _P, _P, _P, _I64, _I32, _I64, _P]),
    "as_fold_ln": (_I32, [_P, _P, _P, _P, _P, _P, _I32, _I32, _I32, _P]),
    "as_attn_softmax": (_I32, [_P, _I64, _I32, _I32, _I32, _I32, _F, _P, _P, _P]),
    "as_embed_posenc": (_I32, [_P, _I64, _P, _P, _P, _I64, _I32, _I32, _P]),
    "as_attn_softmax_bwd": (_I32, [_P, _P, _I64, _I32, _I32, _F, _P]),
    "as_attention_supported": (_I32, [_I32, _I32, _I32, _I32]),
    "as_attention_fwd": (_I32, [_P, _P, _P, _P, _P, _P, _P, _P, _I32, _I32, _I32, _I32, _I32, _I32, _F, _P]),
    "as_attention_fwd_causal": (_I32, [_P, _P, _P, _P, _P, _P, _P, _P, _I32, _I32, _I32, _I32, _I32, _I32, _F, _P]),
    "as_attention_bwd_ds": (_I32, [_P, _P, _P, _P, _P, _I32, _I32, _I32, _I32, _I32, _I32, _F, _P]),
    "as_attention_bwd_ds_causal": (_I32, [_P, _P, _P, _P, _P, _I32, _I32, _I32, _I32, _I32, _I32, _F, _P]),
    "as_attn_softmax_bwd_t": (_I32, [_P, _P, _P, _P, _P, _I32, _I32, _I32, _I32, _I32, _I32, _F, _P]),
    "as_group_reduce": (_I32, [_P, _P, _I32, _I32, _I64, _P, _P]),
    "as_layernorm_bwd": (_I32, [_P, _P, _P, _P, _P, _I64, _I32, _P]),
    "as_unfold_ln": (_I32, [_P, _P, _P, _P, _P, _P, _P, _P, _I32, _I32, _I32, _P]),
    "as_relu_bwd": (_I32, [_P, _P, _P, _I64, _P]),
    "as_add": (_I32, [_P, _P, _P, _I64, _P]),
    "as_row_scale": (_I32, [_P, _P, _P, _I64, _I32, _P]),
    "as_copy_f32": (_I32, [_P, _P, _I64, _P]),
    "as_set_overlap": (None, [_I32]),
    "as_set_matrix_arith": (None, [_I32]),
    "as_get_matrix_arith": (_I32, []),
    "as_conv3x3_stem": (_I32, [_P, _I64, _I64, _I64, _I64, _P, _P, _P, _P, _I32, _I32, _I32, _I32, _P]),
    "as_conv3x3_c32": (_I32, [_P, _P, _P, _P, _P, _I32, _I32, _I32, _P]),
    "as_ln_feat_gelu": (_I32, [_P, _P, _P, _P, _I64, _I32, _I32, _P]),
    "as_gelu": (_I32, [_P, _P, _I64, _P]),
    "as_lstm_bidir_fwd": (_I32, [_P, _P, _I64, _P, _P, _P, _I32, _I32, _I32, _P, _P, _P]),
    "as_lstm_bidir_bwd": (_I32, [_P, _P, _P, _P, _I32, _I32, _I32, _P, _P]),
    "as_gru_unidir_fwd": (_I32, [_P, _P, _P, _P, _I32, _I32, _I32, _P, _P]),
    "as_intersect_semipolar_grid": (_I32, [_P, _P, _I64, _I32, _I32, _I32, _P, _P, _P, _P]),
    "as_artspeech_wait_head_grads": (_I32, [_P, _P]),
    "as_gather_pad_rows": (_I32, [_P, _P, _P, _I32, _I32, _I64, _I32, _D, _P, _P]),
    "as_lin_debug_stamps": (None, [_P, _I64]),
    "as_gru_debug_stamps": (None, [_P]),
    "as_profile_enable": (None, [_I32]),
    "as_profile_reset": (None, []),
    "as_profile_report": (_I32, [C.c_char_p, _I32]),
}

_lib = None


def lib():
    """The loaded library; raises (loudly) if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python -m artspeech_amd.build` "
                "(hipcc --offload-arch=gfx950). artspeech_amd has no CPU fallback.")
        handle = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(handle, name)  # AttributeError if the symbol is not exported
            fn.restype, fn.argtypes = res, args
        _lib = handle
    return _lib


def check(rc, what=""):
    if rc != 0:
        msg = lib().as_last_error().decode()
        raise RuntimeError(f"{what or 'artspeech_hip'} failed (code {rc}): {msg}")


def require_gpu(t, name="tensor"):
    if not t.is_cuda:
        raise RuntimeError(f"artspeech_amd: {name} must live on an MI355X device (got {t.device}); there is no CPU path")
    return t


def stream_ptr():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def layout(dims):
    lay = Layout()
    check(lib().as_artspeech_layout(C.byref(dims), C.byref(lay)), "as_artspeech_layout")
    return lay
