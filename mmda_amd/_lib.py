"""ctypes binding of libmmda_hip.so (the C ABI declared in include/mmda_hip.h).

The product path has NO fallback: if the HIP library is missing or a call fails, this raises.  PyTorch is used by
callers only to own device memory and streams; every pointer handed over here is ``tensor.data_ptr()``.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmmda_hip.so")

F32, BF16 = 0, 1
ACT = {"none": 0, "relu": 1, "sigmoid": 2, "leakyrelu": 3, "tanh": 4, "elu": 5, "hardtanh": 6, "hardshrink": 7, "prelu": 8, "rrelu": 9}

c_f32p = C.c_void_p      # device pointers travel as integers
c_stream = C.c_void_p


class GemmArgs(C.Structure):
    _fields_ = [("mode", C.c_int), ("transA", C.c_int), ("transB", C.c_int), ("M", C.c_int), ("N", C.c_int), ("K", C.c_int),
                ("batch", C.c_int),
                ("A", C.c_void_p), ("lda", C.c_int), ("strideA", C.c_int64),
                ("A2", C.c_void_p), ("gather", C.c_void_p),
                ("B", C.c_void_p), ("ldb", C.c_int), ("strideB", C.c_int64),
                ("C", C.c_void_p), ("ldc", C.c_int), ("strideC", C.c_int64),
                ("bias", C.c_void_p), ("bias2", C.c_void_p), ("strideBias", C.c_int64),
                ("accumulate", C.c_int), ("act", C.c_int),
                ("drop_p", C.c_float), ("drop_seed", C.c_uint64), ("drop_site", C.c_int),
                ("gate", C.c_void_p), ("ldgate", C.c_int), ("gate_scale", C.c_float),
                ("alpha", C.c_float), ("bias_grad", C.c_void_p), ("bias_grad2", C.c_void_p)]


class GemmBf16Args(C.Structure):
    _fields_ = [("M", C.c_int), ("N", C.c_int), ("K", C.c_int), ("A", C.c_void_p), ("lda", C.c_int), ("B", C.c_void_p), ("ldb", C.c_int),
                ("C", C.c_void_p), ("ldc", C.c_int), ("bias", C.c_void_p), ("bias2", C.c_void_p), ("bias_grad", C.c_void_p),
                ("bias_grad2", C.c_void_p), ("accumulate", C.c_int), ("alpha", C.c_float), ("perm_n_H", C.c_int), ("perm_m_H", C.c_int),
                ("tn", C.c_int)]


class ConvertJob(C.Structure):
    _fields_ = [("src", C.c_void_p), ("ld", C.c_int), ("rows", C.c_int), ("cols", C.c_int), ("gather", C.c_void_p),
                ("plain", C.c_void_p), ("ldp", C.c_int), ("transposed", C.c_void_p), ("ldt", C.c_int), ("row_perm_H", C.c_int),
                ("src_bf16", C.c_int)]


class SkinnyArgs(C.Structure):
    _fields_ = [("M", C.c_int), ("N", C.c_int), ("K", C.c_int), ("transB", C.c_int),
                ("A", C.c_void_p), ("A2", C.c_void_p), ("lda", C.c_int), ("B", C.c_void_p), ("ldb", C.c_int),
                ("K2", C.c_int), ("A_2nd", C.c_void_p), ("lda_2nd", C.c_int), ("B_2nd", C.c_void_p), ("ldb_2nd", C.c_int),
                ("C", C.c_void_p), ("ldc", C.c_int), ("C2", C.c_void_p), ("ldc2", C.c_int), ("bias", C.c_void_p),
                ("accumulate", C.c_int), ("alpha", C.c_float), ("act", C.c_int),
                ("drop_p", C.c_float), ("drop_seed", C.c_uint64), ("drop_site", C.c_int),
                ("gate", C.c_void_p), ("ldgate", C.c_int), ("gate_scale", C.c_float),
                ("dsig", C.c_void_p), ("dsig2", C.c_void_p), ("lddsig", C.c_int)]


class TransposeJob(C.Structure):
    _fields_ = [("src", C.c_void_p), ("rows", C.c_int), ("cols", C.c_int), ("ld", C.c_int), ("dst", C.c_void_p), ("ldd", C.c_int)]


class ActParams(C.Structure):
    _fields_ = [("slope", C.c_void_p), ("dslope", C.c_void_p), ("lo", C.c_float), ("hi", C.c_float), ("rand", C.c_int),
                ("seed", C.c_uint64), ("site", C.c_int)]


class LnArgs(C.Structure):
    _fields_ = [("rows", C.c_int), ("n", C.c_int), ("x", C.c_void_p), ("res", C.c_void_p), ("gamma", C.c_void_p),
                ("beta", C.c_void_p), ("y", C.c_void_p), ("mean", C.c_void_p), ("rstd", C.c_void_p), ("act", C.c_int),
                ("drop_p", C.c_float), ("drop_seed", C.c_uint64), ("drop_site", C.c_int),
                ("permute_S", C.c_int), ("permute_B", C.c_int), ("eps", C.c_float), ("actp", ActParams),
                ("y_bf16", C.c_void_p), ("ld_bf16", C.c_int)]


class LnBwdArgs(C.Structure):
    _fields_ = [("rows", C.c_int), ("n", C.c_int), ("dy", C.c_void_p), ("x", C.c_void_p), ("res", C.c_void_p),
                ("gamma", C.c_void_p), ("mean", C.c_void_p), ("rstd", C.c_void_p),
                ("d_x", C.c_void_p), ("accumulate_dx", C.c_int), ("d_res", C.c_void_p),
                ("dgamma", C.c_void_p), ("dbeta", C.c_void_p),
                ("act", C.c_int), ("drop_p", C.c_float), ("drop_seed", C.c_uint64), ("drop_site", C.c_int),
                ("permute_S", C.c_int), ("permute_B", C.c_int), ("actp", ActParams)]


class LstmDesc(C.Structure):
    _fields_ = [("H", C.c_int), ("gates", C.c_void_p), ("cstash", C.c_void_p), ("hseq", C.c_void_p),
                ("wpack", C.c_void_p * 2), ("wpack_c", C.c_void_p * 2), ("utt", C.c_void_p), ("layer", C.c_int), ("d_hseq", C.c_void_p),
                ("xchg", C.c_void_p), ("epoch_base", C.c_uint32), ("gate_minor", C.c_int), ("forward_only", C.c_int),
                ("cell", C.c_int), ("dg_bf16", C.c_void_p), ("dg_bf16_only", C.c_int)]


class GruPadJob(C.Structure):
    _fields_ = [("H", C.c_int), ("D", C.c_int), ("w_ih", C.c_void_p * 2), ("w_hh", C.c_void_p * 2), ("b_ih", C.c_void_p * 2),
                ("b_hh", C.c_void_p * 2), ("pw_ih", C.c_void_p), ("pw_hh", C.c_void_p * 2), ("pb_ih", C.c_void_p), ("pb_hh", C.c_void_p)]


class MisaConfig(C.Structure):
    _fields_ = [("vocab", C.c_int), ("d_t", C.c_int), ("d_v", C.c_int), ("d_a", C.c_int), ("hidden", C.c_int), ("ncls", C.c_int),
                ("act", C.c_int), ("use_cmd_sim", C.c_int), ("use_confidNet", C.c_int),
                ("dropout", C.c_float), ("fusion_dropout", C.c_float), ("threshold", C.c_float),
                ("reverse_grad_weight", C.c_float),
                ("diff_weight", C.c_float), ("sim_weight", C.c_float), ("recon_weight", C.c_float), ("conf_weight", C.c_float),
                ("mode", C.c_int), ("rnncell", C.c_int)]


class Mx8QuantJob(C.Structure):
    _fields_ = [("src", C.c_void_p), ("ld", C.c_int), ("rows", C.c_int), ("K", C.c_int), ("q", C.c_void_p), ("s", C.c_void_p)]


class Mx8Args(C.Structure):
    _fields_ = [("M", C.c_int), ("N", C.c_int), ("K", C.c_int), ("Aq", C.c_void_p), ("As", C.c_void_p), ("Bq", C.c_void_p),
                ("Bs", C.c_void_p), ("C", C.c_void_p), ("ldc", C.c_int), ("bias", C.c_void_p), ("act", C.c_int),
                ("drop_p", C.c_float), ("drop_seed", C.c_uint64), ("drop_site", C.c_int)]


CELL = {"lstm": 0, "gru": 1}

# name -> (restype, argtypes).  Every symbol include/mmda_hip.h declares appears here (tests/test_abi.py checks it).
_P, _I, _I64, _F, _U64 = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_uint64
SIGNATURES = {
    "mmda_last_error": (C.c_char_p, []),
    "mmda_abi_version": (_I, []),
    "mmda_scratch_release": (_I, []),
    "mmda_debug_gemm_dma_mode": (_I, [_I]),
    "mmda_gemm": (_I, [C.POINTER(GemmArgs), _P]),
    "mmda_gemm_grouped": (_I, [C.POINTER(GemmArgs), _I, _P]),
    "mmda_gemm_bf16_grouped": (_I, [C.POINTER(GemmBf16Args), _I, _P]),
    "mmda_convert_bf16": (_I, [C.POINTER(ConvertJob), _I, _P]),
    "mmda_gemm_skinny": (_I, [C.POINTER(SkinnyArgs), _I, _P]),
    "mmda_mx8_quant_bytes": (_I64, [_I, _I]),
    "mmda_mx8_quant": (_I, [C.POINTER(Mx8QuantJob), _I, _P]),
    "mmda_gemm_mx8": (_I, [C.POINTER(Mx8Args), _P]),
    "mmda_transpose_f32": (_I, [C.POINTER(TransposeJob), _I, _P]),
    "mmda_colsum": (_I, [_P, _I, _I, _I, _P, _P, _P]),
    "mmda_embed_gather": (_I, [_P, _P, _I, _I, _P, _P]),
    "mmda_embed_scatter_add": (_I, [_P, _P, _I, _I, _P, _P]),
    "mmda_embed_segment_sum_work_bytes": (_I64, [_I, _I]),
    "mmda_embed_segment_sum": (_I, [_P, _P, _I, _I, _P, _P, _I64, _P]),
    "mmda_allreduce": (_I, [_P, C.c_size_t, _P, _P]),
    "mmda_layernorm_fwd": (_I, [C.POINTER(LnArgs), _P]),
    "mmda_layernorm_bwd": (_I, [C.POINTER(LnBwdArgs), _P]),
    "mmda_layernorm_fwd_multi": (_I, [C.POINTER(LnArgs), _I, _P]),
    "mmda_layernorm_bwd_multi": (_I, [C.POINTER(LnBwdArgs), _I, _P]),
    "mmda_layernorm_param_grads": (_I, [C.POINTER(LnBwdArgs), _I, _P]),
    "mmda_lstm_packed_bytes": (_I64, [_I, _I, _I]),
    "mmda_lstm_xchg_bytes": (_I64, [_I, _I]),
    "mmda_lstm_resident_applicable": (_I, [_I, _I, C.POINTER(LstmDesc), _I, _I, _I]),
    "mmda_lstm_bwd_emits_dg_bf16": (_I, [_I, _I, C.POINTER(LstmDesc), _I, _I]),
    "mmda_debug_set_lstm_stamps": (_I, [_P]),
    "mmda_lstm_pack_whh": (_I, [_I, _I, _P, _P, _P, _P]),
    "mmda_lstm_pack_whh_multi": (_I, [_I, _I, C.POINTER(C.c_int), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                                      C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), _P]),
    "mmda_lstm_pack_whh_cluster": (_I, [_I, _P, _P, _P]),
    "mmda_lstm_pack_whh_and_convert": (_I, [_I, C.POINTER(C.c_int), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                                            C.POINTER(C.c_void_p), C.POINTER(ConvertJob), _I, _P]),
    "mmda_lstm_pack_convert_transpose": (_I, [_I, C.POINTER(C.c_int), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                                              C.POINTER(C.c_void_p), C.POINTER(ConvertJob), _I, C.POINTER(TransposeJob), _I, _P]),
    "mmda_lstm_fwd": (_I, [_I, _I, C.POINTER(LstmDesc), _I, _I, _P, _P]),
    "mmda_lstm_bwd": (_I, [_I, _I, C.POINTER(LstmDesc), _I, _I, _P, _P]),
    "mmda_gru_pad_params": (_I, [C.POINTER(GruPadJob), _I, _P]),
    "mmda_gru_unpad_grads": (_I, [C.POINTER(GruPadJob), _I, _P]),
    "mmda_attn_fwd": (_I, [_P, _I, _I, _I, _I, _P, _P, _F, _U64, _I, _P]),
    "mmda_attn_bwd": (_I, [_P, _P, _P, _I, _I, _I, _I, _P, _F, _U64, _I, _P]),
    "mmda_add": (_I, [_P, _P, _P, _I64, _P]),
    "mmda_sigmoid_bwd_inplace": (_I, [_P, _P, _I64, _P]),
    "mmda_act_dropout_fwd": (_I, [_P, _P, _I64, _I, _F, _U64, _I, _P]),
    "mmda_act_dropout_bwd": (_I, [_P, _P, _P, _I64, _I, _F, _U64, _I, _P]),
    "mmda_act_dropout_fwd_p": (_I, [_P, _P, _I64, _I, C.POINTER(ActParams), _F, _U64, _I, _P]),
    "mmda_act_dropout_bwd_p": (_I, [_P, _P, _P, _I64, _I, C.POINTER(ActParams), _F, _U64, _I, _P]),
    "mmda_heads_fwd": (_I, [_P, _I, _I, _F, _P, _P, _P, _F, _U64, _I, _P]),
    "mmda_heads_bwd": (_I, [_P, _P, _P, _P, _I, _I, _P, _F, _U64, _I, _P]),
    "mmda_loss_cls": (_I, [_P, _P, _I, _I, _F, _P, _P, _P]),
    "mmda_loss_conf": (_I, [_P, _P, _P, _I, _I, _F, _P, _P, _P, _P]),
    "mmda_loss_diff": (_I, [_P, _I64, _I, _I, _F, _P, _P, _P, _P]),
    "mmda_loss_diff_work_floats": (_I64, [_I, _I]),
    "mmda_loss_diff_pairs": (_I, [_P, _I64, _I, _I, C.POINTER(C.c_int), _I, _I, _F, _P, _P, _P, _P]),
    "mmda_loss_cmd_pairs": (_I, [_P, _I64, _I, _I, C.POINTER(C.c_int), _I, _I, _I, _F, _F, _P, _P, _P]),
    "mmda_loss_cmd": (_I, [_P, _I64, _I, _I, _F, _P, _P, _P]),
    "mmda_loss_recon": (_I, [_P, _P, _I64, _I, _I, _F, _P, _P, _P, _P]),
    "mmda_loss_misc": (_I, [_P, _P, _P, _I, _I, _P, _P, _I, _I, _F, _P, _P, _I64, _F, _P, _P, _P, _F, _F, _F, _F, _I, _P]),
    "mmda_eval_accumulate": (_I, [_P, _P, _I, _I, _P, _P]),
    "mmda_loss_domain": (_I, [_P, _I, _F, _P, _P, _P]),
    "mmda_clamp_adam": (_I, [_P, _P, _P, _P, _I64, _F, _F, _F, _F, _F, _F, _I, _P]),
    "mmda_clamp_adam_rows": (_I, [_P, _P, _P, _P, _I, _I, _P, _I, _F, _F, _F, _F, _F, _F, _I, _P]),
    "mmda_mark_rows": (_I, [_P, _I, _P, _I, _P]),
    "mmda_clamp": (_I, [_P, _I64, _F, _P]),
    "mmda_clamp_rmsprop": (_I, [_P, _P, _P, _I64, _F, _F, _F, _F, _F, _P]),
    "mmda_misa_create": (_I, [C.POINTER(MisaConfig), C.POINTER(C.c_void_p)]),
    "mmda_misa_destroy": (None, [_P]),
    "mmda_misa_num_params": (_I, [_P]),
    "mmda_misa_param_info": (_I, [_P, _I, C.POINTER(C.c_char_p), C.POINTER(_I64), C.POINTER(_I), C.POINTER(_I)]),
    "mmda_misa_flat_floats": (_I64, [_P]),
    "mmda_misa_dense_floats": (_I64, [_P]),
    "mmda_misa_bind": (_I, [_P, _P, _P, _P, _P]),
    "mmda_misa_workspace_floats": (_I64, [_P, _I, _I]),
    "mmda_misa_set_workspace": (_I, [_P, _P, _I64, _I, _I]),
    "mmda_misa_set_workspace_async": (_I, [_P, _P, _I64, _I, _I, _P]),
    "mmda_misa_tensor_offset": (_I64, [_P, C.c_char_p]),
    "mmda_misa_set_mode": (_I, [_P, _I]),
    "mmda_misa_set_overlap": (_I, [_P, _I]),
    "mmda_misa_set_recurrence": (_I, [_P, _I]),
    "mmda_misa_set_gemm_operands": (_I, [_P, _I]),
    "mmda_misa_early_grad_floats": (_I64, [_P]),
    "mmda_misa_wait_early_grads": (_I, [_P, _P]),
    "mmda_misa_set_inference": (_I, [_P, _I]),
    "mmda_misa_set_fusion_fp8": (_I, [_P, _I]),
    "mmda_misa_cluster_status": (_I, [_P, C.POINTER(_I)]),
    "mmda_misa_forward": (_I, [_P, _P, _P, _P, _P, _I, _U64, _P]),
    "mmda_misa_losses": (_I, [_P, _P, _I, _P]),
    "mmda_misa_set_external_batch_losses": (_I, [_P, _I]),
    "mmda_misa_backward": (_I, [_P, _P, _P, _P, _P, _P]),
    "mmda_misa_zero_grad": (_I, [_P, _P]),
    "mmda_misa_zero_act_grads": (_I, [_P, _P]),
    "mmda_misa_adam_step": (_I, [_P, _F, _F, _F, _I, _P]),
    "mmda_misa_timing_stride": (_I, [_P, _I]),
    "mmda_misa_timing_rotate": (_I, [_P, _I]),
    "mmda_misa_timing_begin": (_I, [_P, _I]),
    "mmda_misa_timing_collect": (_I, [_P, C.POINTER(C.c_float * 4), C.POINTER(_I)]),
    "mmda_misa_timing_end": (_I, [_P]),
    "mmda_misa_train_step": (_I, [_P, _P, _P, _P, _P, _P, _I, _U64, _I, _F, _F, _I, _P]),
}

_lib = None


class MMDAError(RuntimeError):
    pass


def load():
    """Load the library once.  Raises (never falls back) if it is missing: build it with
    ``python -c 'import __graft_entry__ as g; g.build()'`` or ``make -C mmda_amd/csrc``."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MMDAError(f"{LIB_PATH} not found: the HIP hot-path library is not built (make -C mmda_amd/csrc); "
                        "there is no CPU/PyTorch fallback")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError here = header/binding drift
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, what=""):
    if rc != 0:
        lib = load()
        msg = lib.mmda_last_error()
        raise MMDAError(f"{what} failed with code {rc}: {msg.decode() if msg else ''}")


def ptr(t):
    """Device pointer of a torch tensor (or None)."""
    return None if t is None else t.data_ptr()


def stream_ptr():
    import torch
    return torch.cuda.current_stream().cuda_stream
