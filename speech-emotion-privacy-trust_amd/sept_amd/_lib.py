"""ctypes binding of libsept_hip.so (C ABI in include/sept.h).  Fails loudly when the
library has not been built: the product path never falls back to CPU/eager code."""
import ctypes
import os
from ctypes import POINTER, c_char_p, c_double, c_float, c_int, c_long, c_size_t, c_ulonglong, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "..", "csrc", "libsept_hip.so")


class SeptError(RuntimeError):
    """A libsept_hip entry point returned a negative status."""


def _load():
    path = os.path.abspath(os.environ.get("SEPT_LIB", LIB_PATH))   # SEPT_LIB: another build of the same library (A/B aid)
    if not os.path.exists(path):
        raise ImportError(
            f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C speech-emotion-privacy-trust_amd/csrc` (there is no CPU fallback)")
    return ctypes.CDLL(path)


lib = _load()


class PrepItem(ctypes.Structure):
    """sept_prep_item of include/sept.h (one weight-only operand build of sept_prepare_operands)"""
    _fields_ = [("kind", c_int), ("src0", c_void_p), ("src1", c_void_p), ("src2", c_void_p), ("src3", c_void_p),
                ("dst0", c_void_p), ("dst1", c_void_p), ("dst2", c_void_p), ("p0", c_int), ("p1", c_int), ("p2", c_int),
                ("p3", c_int)]


PREP_CONV1, PREP_CONV5X5, PREP_GRU = 1, 2, 3

# name -> (restype, argtypes); kept in one table so tests can check it against sept.h
SIGNATURES = {
    "sept_last_error": (c_char_p, []),
    "sept_abi_version": (c_int, []),
    "sept_device_check": (c_int, []),
    "sept_mel_plan_create": (c_int, [c_int, c_int, c_int, POINTER(c_float), POINTER(c_float), POINTER(c_void_p)]),
    "sept_mel_plan_destroy": (c_int, [c_void_p]),
    "sept_mel_num_frames": (c_int, [c_void_p, c_int]),
    "sept_mel_forward": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_int, c_void_p]),
    "sept_mel_kernel_name": (c_char_p, [c_void_p]),
    "sept_conv5x5_prep_weights": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "sept_prepare_operands": (c_int, [POINTER(PrepItem), c_int, c_void_p]),
    "sept_conv5x5_forward": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
                                     c_void_p]),
    "sept_conv5x5_stats_parts": (c_int, [c_int] * 5),
    "sept_conv5x5_forward_stats": (c_int, [c_void_p] * 5 + [c_int] * 5 + [c_void_p]),
    "sept_conv5x5_wgrad_workspace_floats": (c_size_t, [c_int, c_int]),
    "sept_conv5x5_backward_weight": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
                                             c_void_p]),
    "sept_conv1_prep_floats": (c_size_t, []),
    "sept_conv1_forward": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "sept_conv1_backward_data": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "sept_conv1_workspace_floats": (c_size_t, []),
    "sept_conv1_backward_weight": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                                           c_void_p]),
    "sept_bn_workspace_floats": (c_size_t, [c_int]),
    "sept_bn_stats": (c_int, [c_void_p, c_long, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                              c_float, c_float, c_void_p]),
    "sept_bn_eval_stats": (c_int, [c_void_p, c_void_p, c_int, c_float, c_void_p, c_void_p, c_void_p]),
    "sept_bn_relu_pool_forward": (c_int, [c_void_p] * 7 + [c_int] * 5 + [c_void_p]),
    "sept_bn_relu_pool_backward": (c_int, [c_void_p] * 12 + [c_int] * 5 + [c_void_p]),
    "sept_gemm_nt_split": (c_int, [c_void_p, c_long, c_int, c_void_p, c_long, c_void_p, c_long, c_int, c_void_p,
                                   c_int, c_int, c_int, c_void_p]),
    "sept_gemm_tn_workspace_floats": (c_size_t, [c_int, c_int]),
    "sept_gemm_tn_split": (c_int, [c_void_p, c_long, c_void_p, c_long, c_int, c_void_p, c_long, c_int, c_int, c_int,
                                   c_void_p, c_long, c_void_p]),
    "sept_gemm": (c_int, [c_void_p, c_long, c_long, c_int, c_void_p, c_long, c_long, c_int, c_void_p, c_long, c_int,
                          c_void_p, c_int, c_int, c_int, c_float, c_float, c_void_p, c_long, c_void_p]),
    "sept_gru_forward": (c_int, [c_void_p] * 7 + [c_int] * 3 + [c_void_p]),
    "sept_gru_backward": (c_int, [c_void_p] * 8 + [c_int] * 3 + [c_void_p]),
    "sept_cloak_forward": (c_int, [c_void_p] * 5 + [c_float, c_float, c_void_p, c_int, c_long, c_void_p]),
    "sept_cloak_forward_rows": (c_int, [c_void_p] * 4 + [c_int, c_void_p, c_float, c_float, c_void_p, c_int, c_long,
                                        c_void_p]),
    "sept_cloak_scales": (c_int, [c_void_p, c_float, c_float, c_void_p, c_void_p, c_long, c_void_p]),
    "sept_cloak_backward": (c_int, [c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_void_p, c_float, c_float,
                                    c_float, c_void_p, c_void_p, c_void_p, c_int, c_long, c_void_p]),
    "sept_scale": (c_int, [c_void_p, c_float, c_void_p, c_long, c_void_p]),
    "sept_fill": (c_int, [c_void_p, c_float, c_long, c_void_p]),
    "sept_mul": (c_int, [c_void_p, c_void_p, c_void_p, c_long, c_void_p]),
    "sept_debug_stamp": (c_int, [c_void_p, c_void_p]),
    "sept_kclock_next": (c_int, [c_void_p]),
    "sept_window_norm_cloak": (c_int, [c_void_p] * 7 + [c_float, c_float, c_void_p] + [c_int] * 6 + [c_long, c_void_p]),
    "sept_add": (c_int, [c_void_p, c_void_p, c_void_p, c_long, c_void_p]),
    "sept_relu_dropout_forward": (c_int, [c_void_p, c_void_p, c_void_p, c_long, c_void_p]),
    "sept_relu_dropout_backward": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_long, c_void_p]),
    "sept_mean_t_forward": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "sept_mean_t_backward": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "sept_colsum_workspace_floats": (c_size_t, [c_int]),
    "sept_colsum": (c_int, [c_void_p, c_long, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p]),
    "sept_cross_entropy": (c_int, [c_void_p, c_void_p, c_void_p, c_float, c_int, c_int, c_void_p, c_void_p, c_int,
                                   c_void_p]),
    "sept_softmax_mean": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "sept_loss_sub_log": (c_int, [c_void_p, c_void_p, c_float, c_void_p]),
    "sept_permute_cols": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "sept_unfold1d": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "sept_fold1d": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "sept_relu_pool1d_forward": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "sept_relu_pool1d_backward": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                          c_void_p]),
    "sept_dropout_mask": (c_int, [c_void_p, c_long, c_float, c_ulonglong, c_void_p, c_ulonglong, c_void_p]),
    "sept_normal": (c_int, [c_void_p, c_long, c_float, c_float, c_ulonglong, c_void_p, c_ulonglong, c_void_p]),
    "sept_counter_add": (c_int, [c_void_p, c_long, c_void_p]),
    "sept_tanh_forward": (c_int, [c_void_p, c_void_p, c_long, c_void_p]),
    "sept_tanh_backward": (c_int, [c_void_p, c_void_p, c_void_p, c_long, c_void_p]),
    "sept_att_pool_forward": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "sept_att_pool_backward": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                       c_void_p]),
    "sept_speaker_stats_workspace_doubles": (c_size_t, [c_int, c_int]),
    "sept_speaker_stats": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "sept_speaker_stats_windows": (c_int, [c_void_p] * 4 + [c_int] * 6 + [c_void_p, c_void_p, c_void_p]),
    "sept_window_norm_spk": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int,
                                     c_int, c_void_p]),
    "sept_add_normal": (c_int, [c_void_p, c_void_p, c_long, c_float, c_ulonglong, c_void_p, c_ulonglong, c_void_p]),
    "sept_resample_forward": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_long, c_int, c_int, c_int, c_long, c_void_p]),
    "sept_bn_partial_sums": (c_int, [c_void_p, c_long, c_int, c_void_p, c_void_p, c_void_p]),
    "sept_bn_stats_from_sums": (c_int, [c_void_p, c_double, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                        c_float, c_float, c_void_p]),
    "sept_bn_relu_pool_backward_reduce": (c_int, [c_void_p] * 12 + [c_int] * 5 + [c_void_p]),
    "sept_bn_relu_pool_backward_apply": (c_int, [c_void_p] * 8 + [c_double, c_void_p] + [c_int] * 5 + [c_void_p]),
    "sept_conv1_stats_parts": (c_int, [c_int, c_int]),
    "sept_conv1_forward_stats": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                                         c_void_p]),
    "sept_bn_stats_from_partials": (c_int, [c_void_p, c_int, c_long, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                            c_void_p, c_float, c_float, c_void_p]),
    "sept_lstm_forward": (c_int, [c_void_p] * 8 + [c_int, c_int, c_int, c_void_p]),
    "sept_lstm_backward": (c_int, [c_void_p] * 8 + [c_int, c_int, c_int, c_void_p]),
    "sept_head_forward": (c_int, [c_void_p] * 10 + [c_int] * 5 + [c_void_p]),
    "sept_head_backward": (c_int, [c_void_p] * 7 + [c_int] * 5 + [c_void_p]),
    "sept_head_backward_ce": (c_int, [c_void_p] * 3 + [c_float] + [c_void_p] * 7 + [c_int] * 5 + [c_void_p]),
    "sept_scale_dev": (c_int, [c_void_p, c_void_p, c_void_p, c_long, c_void_p]),
    "sept_gru_pack": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p,
                              c_void_p, c_void_p]),
    "sept_topdb_clamp": (c_int, [c_void_p, c_int, c_long, c_float, c_void_p]),
    "sept_gradient1d": (c_int, [c_void_p, c_void_p, c_int, c_long, c_float, c_void_p]),
    "sept_transpose_last2": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "sept_window_norm": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int,
                                 c_void_p]),
    "sept_sgd_step": (c_int, [c_void_p, c_void_p, c_void_p, c_long, c_float, c_float, c_float, c_int, c_float,
                              c_void_p]),
    "sept_adam_step": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_long, c_float, c_float, c_float, c_float,
                               c_float, c_int, c_float, c_void_p]),
    "sept_conv5x5_bwsums_parts": (c_int, [c_int] * 5),
    "sept_conv5x5_variant": (c_int, [c_int] * 4 + [POINTER(c_int)]),
    "sept_conv5x5_bnapply_parts": (c_int, [c_int] * 6),
    "sept_conv5x5_act_parts": (c_int, [c_int] * 6),
    "sept_conv5x5_forward_act": (c_int, [c_void_p] * 10 + [c_int] * 5 + [c_void_p]),
    "sept_conv5x5_dgrad_bnapply": (c_int, [c_void_p] * 17 + [c_int] * 5 + [c_void_p]),
    "sept_conv5x5_dgrad_bnsums": (c_int, [c_void_p] * 8 + [c_int] * 5 + [c_void_p]),
    "sept_bn_relu_pool_backward_presummed": (c_int, [c_void_p] * 8 + [c_int, c_void_p, c_void_p, c_void_p, c_void_p] +
                                             [c_int] * 5 + [c_void_p]),
    "sept_bn_backward_sums_presummed": (c_int, [c_void_p] * 8 + [c_int, c_void_p, c_void_p, c_void_p, c_void_p] + [c_int] * 5 +
                                        [c_void_p]),
    "sept_bn_relu_pool_forward_argmax": (c_int, [c_void_p] * 8 + [c_int] * 5 + [c_void_p]),
    "sept_conv1_backward_data_sparse": (c_int, [c_void_p] * 10 + [c_double] + [c_void_p] * 4 + [c_int] * 3 + [c_void_p]),
    "sept_conv1_prep": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p]),
    "sept_conv1_fused_supported": (c_int, [c_int, c_int]),
    "sept_conv1_bn_relu_pool_forward": (c_int, [c_void_p] * 10 + [c_int] * 3 + [c_void_p]),
    "sept_bn_bwd_sums_from_partials": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "sept_conv1_pool_supported": (c_int, [c_int, c_int]),
    "sept_conv1_pool_backward_supported": (c_int, [c_int, c_int]),
    "sept_conv1_coef_floats": (c_size_t, []),
    "sept_conv1_forward_pool": (c_int, [c_void_p] * 8 + [c_int] * 3 + [c_void_p]),
    "sept_bn_relu_ext_forward": (c_int, [c_void_p] * 8 + [c_int, c_long, c_int, c_void_p]),
    "sept_bn_backward_sums_ext": (c_int, [c_void_p] * 11 + [c_int, c_long, c_int, c_void_p]),
    "sept_conv5x5_dgrad_bnsums_ext": (c_int, [c_void_p] * 10 + [c_int] * 5 + [c_void_p]),
    "sept_conv1_wgrad_sparse_workspace_floats": (c_size_t, []),
    "sept_conv1_backward_weight_sparse": (c_int, [c_void_p] * 10 + [c_double] + [c_void_p] * 3 + [c_int] * 3 + [c_void_p]),
    "sept_conv1_dsum_workspace_floats": (c_size_t, [c_int, c_int]),
    "sept_conv1_backward_data_sum": (c_int, [c_void_p] * 10 + [c_double] + [c_void_p] * 3 + [c_int] * 3 + [c_void_p]),
    "sept_copy_bytes": (c_int, [c_void_p, c_void_p, c_long, c_void_p]),
    "sept_gru_forward_masked": (c_int, [c_void_p] * 9 + [c_int] * 3 + [c_void_p]),
    "sept_gru_backward_masked": (c_int, [c_void_p] * 9 + [c_int] * 3 + [c_void_p]),
    "sept_counter_add2": (c_int, [c_void_p, c_void_p, c_long, c_void_p]),
    "sept_gemm_tn_split_colsum": (c_int, [c_void_p, c_long, c_void_p, c_long, c_int, c_void_p, c_long, c_void_p, c_int, c_int,
                                          c_int, c_void_p, c_long, c_void_p]),
    "sept_sgd_step_dev": (c_int, [c_void_p, c_void_p, c_void_p, c_long, c_void_p, c_float, c_float, c_float, c_void_p]),
    "sept_adam_step_dev": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_long, c_void_p, c_float, c_float, c_float,
                                   c_float, c_void_p, c_float, c_void_p]),
}

for _name, (_res, _args) in SIGNATURES.items():
    _fn = getattr(lib, _name)
    _fn.restype = _res
    _fn.argtypes = _args


def check(status: int, what: str = "") -> int:
    if status < 0:
        msg = lib.sept_last_error()
        raise SeptError(f"{what or 'libsept_hip'} failed with status {status}: "
                        f"{msg.decode() if msg else '?'}")
    return status


def require_cuda(*tensors):
    """Every op runs on the GPU through the HIP library; anything else is an error."""
    import torch
    for t in tensors:
        if not isinstance(t, torch.Tensor):
            raise TypeError(f"expected a torch.Tensor, got {type(t)}")
        if not t.is_cuda:
            raise SeptError("libsept_hip ops need CUDA(HIP) tensors: there is no CPU fallback "
                            f"(got a tensor on {t.device})")


def current_stream_ptr(device=None) -> int:
    import torch
    return torch.cuda.current_stream(device).cuda_stream
