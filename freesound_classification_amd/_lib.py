"""ctypes binding of libfsc_hip.so (the C ABI declared in include/fsc_hip.h).

There is deliberately no fallback: if the shared library is missing or a call fails, the
product path raises.  Tensors cross the boundary as raw device pointers
(``tensor.data_ptr()``) plus explicit sizes; the stream is torch's current HIP stream.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libfsc_hip.so")

_lib = None


class FscError(RuntimeError):
    pass


class ConvDesc(C.Structure):
    _fields_ = [("n", C.c_int), ("c_in", C.c_int), ("c_out", C.c_int), ("h", C.c_int),
                ("w", C.c_int), ("kh", C.c_int), ("kw", C.c_int), ("arith", C.c_int)]

    def __init__(self, n, c_in, c_out, h, w, kh, kw, arith=-1):      # -1 = FSC_ARITH_DEFAULT
        super().__init__(n, c_in, c_out, h, w, kh, kw, arith)


class OptTensor(C.Structure):
    _fields_ = [("param", C.c_void_p), ("grad", C.c_void_p), ("state0", C.c_void_p),
                ("state1", C.c_void_p), ("state2", C.c_void_p), ("count", C.c_long)]


_P, _I, _L, _F, _SZ = C.c_void_p, C.c_int, C.c_long, C.c_float, C.c_size_t
_U64 = C.c_uint64
_D = C.POINTER(ConvDesc)

# name -> (restype, argtypes); restype int means "0 on success"
SIGNATURES = {
    "fsc_version": (_I, []),
    "fsc_last_error_string": (C.c_char_p, []),
    "fsc_frontend_table_floats": (_SZ, [_I]),
    "fsc_frontend_tables_init": (_I, [_P, _I, _P]),
    "fsc_frontend_logmel_fwd": (_I, [_P, _I, _I, _L, _I, _I, _P, _P, _P, _P, _I, _I, _F, _P, _L, _I, _P]),
    "fsc_frontend_stft_fwd": (_I, [_P, _I, _I, _L, _I, _I, _P, _I, _F, _P, _L, _I, _P]),
    "fsc_conv_packed_floats": (_SZ, [_D, _I]),
    "fsc_conv_pack_weights": (_I, [_D, _P, _I, _P, _P]),
    "fsc_conv_pack_weights_multi_supported": (_I, [_D, _I]),
    "fsc_conv_pack_weights_multi": (_I, [_I, _P, _P, _P, _P, _P]),
    "fsc_conv_fwd": (_I, [_D, _P, _P, _P, _I, _I, _P, _P, _P]),
    "fsc_amax": (_I, [_P, _L, _P, _P]),
    "fsc_conv_pool_supported": (_I, [_D]),
    "fsc_conv_pool_fwd": (_I, [_D, _P, _P, _P, _P, _P, _P]),
    "fsc_conv_stem_wgrad_pooled_blocks": (_SZ, [_D]),
    "fsc_conv_stem_wgrad_pooled": (_I, [_D, _P, _P, _P, _P, _P, _P, _P]),
    "fsc_conv_stem_grads_finish": (_I, [_D, _P, _P, _P, _P, _P, _I, _P, _P, _P, _P, _P]),
    "fsc_absmin": (_I, [_P, _L, _P, _P]),
    "fsc_cat_cols": (_I, [_P, _P, _I, _I, _P, _I, _P]),
    "fsc_bump_counters": (_I, [_P, _I, _P]),
    "fsc_conv_l16_last_clock": (_I, [_I, C.POINTER(C.c_double)]),
    "fsc_conv_l16_pack_weights_multi": (_I, [_I, _P, _P, _P, _P, _P]),
    "fsc_bn_train_stats_conv": (_I, [_P, _I, _I, _I, _I, _P, _I, _I, _L, _P, _P, _F, _F, _P, _P, _P, _P, _P, _P, _P, _P]),
    "fsc_conv_plan_describe": (_I, [_D, _I, C.c_char_p, _SZ]),
    "fsc_conv_default_arith": (_I, []),
    "fsc_conv_wgrad_workspace_bytes": (_SZ, [_D]),
    "fsc_conv_wgrad": (_I, [_D, _P, _P, _P, _P, _P, _P, _P]),
    "fsc_conv_wgrad_partial": (_I, [_D, _P, _P, _P, _P, _P, _P]),
    "fsc_conv_wgrad_reduce_multi": (_I, [_I, _P, _P, _P, _P]),
    "fsc_l16_bytes": (_SZ, [_I, _I, _L]),
    "fsc_l16_pack": (_I, [_P, _I, _I, _L, _P, _P, _P]),
    "fsc_l16_unpack": (_I, [_P, _I, _I, _L, _P, _P, _P]),
    "fsc_l16_bytes_limbs": (_SZ, [_I, _I, _L, _I]),
    "fsc_l16_pack_limbs": (_I, [_P, _I, _I, _L, _P, _I, _P, _P]),
    "fsc_l16_unpack_limbs": (_I, [_P, _I, _I, _L, _P, _I, _P, _P]),
    "fsc_conv_l16_supported": (_I, [_D, _I]),
    "fsc_conv_l16_packed_floats": (_SZ, [_D, _I]),
    "fsc_conv_l16_pack_weights": (_I, [_D, _P, _I, _P, _P]),
    "fsc_conv_l16_pack_weights_pair": (_I, [_D, _P, _P, _P, _P]),
    "fsc_conv_l16_fwd": (_I, [_D, _P, _P, _P, _P, _I, _I, _P, _P]),
    "fsc_conv_l16_plan_describe": (_I, [_D, _I, C.c_char_p, _SZ]),
    "fsc_conv_l16_fwd_act_supported": (_I, [_D]),
    "fsc_conv_l16_fwd_act": (_I, [_D, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "fsc_conv_l16_pool_fwd_act_supported": (_I, [_D]),
    "fsc_conv_l16_pool_fwd_act": (_I, [_D, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "fsc_conv_l16_pool_supported": (_I, [_D]),
    "fsc_conv_l16_pool_fwd": (_I, [_D, _P, _P, _P, _P, _P, _P, _P]),
    "fsc_conv_l16_wgrad_supported": (_I, [_D]),
    "fsc_conv_l16_wgrad_workspace_bytes": (_SZ, [_D]),
    "fsc_conv_l16_wgrad": (_I, [_D, _P, _P, _P, _P, _P, _P, _P]),
    "fsc_conv_l16_wgrad_plan_describe": (_I, [_D, C.c_char_p, _SZ]),
    "fsc_bn_workspace_bytes": (_SZ, [_I]),
    "fsc_bn_workspace_ticket_offset": (_SZ, [_I]),
    "fsc_bn_workspace_reset": (_I, [_P, _I, _P]),
    "fsc_bn_train_stats": (_I, [_P, _I, _I, _L, _P, _P, _F, _F, _P, _P, _P, _P, _P, _P, _P, _P, _I, _P, _P]),
    "fsc_bn_train_act_fwd_supported": (_I, [_I, _I, _L]),
    "fsc_bn_train_act_fwd": (_I, [_P, _P, _I, _I, _L, _P, _P, _F, _F, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "fsc_bn_eval_prepare": (_I, [_I, _P, _P, _P, _P, _F, _P, _P, _P]),
    "fsc_conv_l16_stats_layout": (_I, [_D, _I, _P]),
    "fsc_conv_fwd_stats_layout": (_I, [_D, _P]),
    "fsc_conv_fwd_stats": (_I, [_D, _P, _P, _P, _P, _P, _P, _P]),
    "fsc_conv_fwd_pool_stats_supported": (_I, [_D]),
    "fsc_conv_fwd_pool_stats": (_I, [_D, _P, _P, _P, _P, _P, _P, _P, _P]),
    "fsc_conv_l16_fwd_stats": (_I, [_D, _P, _P, _P, _P, _P, _P, _P, _P]),
    "fsc_conv_l16_pool_fwd_stats": (_I, [_D, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "fsc_bn_records_fold_conv": (_I, [_P, _I, _I, _I, _I, _I, _P, _P]),
    "fsc_bn_records_bytes": (_SZ, [_I, _I, _L]),
    "fsc_bn_act_fwd_rec": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _L, _P, _P]),
    "fsc_bn_records_fold": (_I, [_P, _P, _I, _I, _L, _P, _P, _P, _P]),
    "fsc_bn_act_fwd": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _L, _P, _P, _P, _P]),
    "fsc_bn_act_fwd_limbs": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _L, _P, _P, _P, _I, _P]),
    "fsc_bn_act_bwd": (_I, [_P] * 16 + [_I, _I, _L, _P, _P, _P, _I, _P, _P]),
    "fsc_bn_act_bwd_unpool": (_I, [_P] * 13 + [_I, _I, _I, _I, _I, _P, _P, _P, _I, _P, _P]),
    "fsc_maxpool_fwd": (_I, [_P, _P, _P, _I, _I, _I, _I, _P]),
    "fsc_maxpool_bwd": (_I, [_P, _P, _P, _I, _I, _I, _I, _P]),
    "fsc_global_maxpool_fwd": (_I, [_P, _P, _P, _I, _L, _P]),
    "fsc_global_maxpool_bwd": (_I, [_P, _P, _P, _P, _I, _L, _P]),
    "fsc_linear_fwd": (_I, [_P, _P, _P, _P, _I, _I, _I, _P]),
    "fsc_linear_bwd": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "fsc_dropout_fwd": (_I, [_P, _P, _P, _L, _F, _U64, _U64, _P]),
    "fsc_dropout_bwd": (_I, [_P, _P, _P, _L, _F, _P]),
    "fsc_lsep_fwd": (_I, [_P, _P, _P, _I, _I, _P]),
    "fsc_lsep_bwd": (_I, [_P, _P, _P, _P, _I, _I, _P]),
    "fsc_bce_fwd": (_I, [_P, _P, _P, _P, _L, _P]),
    "fsc_bce_bwd": (_I, [_P, _P, _P, _P, _L, _P]),
    "fsc_sigmoid": (_I, [_P, _P, _L, _P]),
    "fsc_mean_fwd": (_I, [_P, _P, _L, _F, _P]),
    "fsc_mean_bwd": (_I, [_P, _P, _L, _F, _P]),
    "fsc_mixup_batch": (_I, [_P] * 8 + [_I, _L, _L, _L, _P, _P, _P, _I, _P]),
    "fsc_mixup_rows": (_I, [_P] * 9 + [_I, _L, _L, _L, _P, _P, _P, _I, _P]),
    "fsc_segments_gather": (_I, [_P, _L, _P, _P, _P, _P, _I, _P, _I, _L, _P]),
    "fsc_freq_mean_fwd": (_I, [_P, _P, _I, _I, _I, _I, _P]),
    "fsc_freq_mean_bwd": (_I, [_P, _P, _I, _I, _I, _I, _P]),
    "fsc_layernorm_fwd": (_I, [_P, _P, _P, _F, _P, _P, _P, _L, _I, _P]),
    "fsc_layernorm_bwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _L, _I, _P]),
    "fsc_gru_step_fwd": (_I, [_P, _L, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _P]),
    "fsc_gru_step_bwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _L, _P, _P, _I, _I, _P]),
    "fsc_adam_amsgrad_step": (_I, [C.POINTER(OptTensor), _I, _F, _F, _F, _F, _F, _I, _F, _P]),
    "fsc_adam_step_factors": (None, [_F, _F, _F, _I, C.POINTER(C.c_float)]),
    "fsc_adam_amsgrad_step_dev": (_I, [C.POINTER(OptTensor), _I, _P, _F, _F, _F, _F, _F, _P]),
    "fsc_sgd_nesterov_step": (_I, [C.POINTER(OptTensor), _I, _F, _F, _F, _I, _F, _P]),
    "fsc_plane_border_sums": (_I, [_P, _I, _I, _I, _I, _P, _P]),
    "fsc_first_block_1d_finish": (_I, [_P] * 8 + [_I, _I, _I, _I, _P, _P, _P, _P]),
    "fsc_fill": (_I, [_P, _F, _L, _P]),
    "fsc_axpy": (_I, [_P, _F, _P, _L, _P]),
}


def load():
    """Load libfsc_hip.so once; raise if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise FscError(
            "libfsc_hip.so is not built (expected at %s). Run `python -c 'import __graft_entry__ "
            "as g; g.build()'` or `make -C freesound_classification_amd/csrc`. There is no CPU "
            "fallback for the accelerated path." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError here = header/library mismatch
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def exported_symbols():
    return sorted(SIGNATURES)


_get_device = torch._C._cuda_getDevice if hasattr(torch._C, "_cuda_getDevice") else torch.cuda.current_device
_get_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def stream_ptr():
    """Raw handle of torch's current stream on the current device.  (torch.cuda.current_stream() builds a Stream object through
    three Python layers: 8.5 us a call, 3.6 ms of a 12.8 ms launch-bound cfg-3 step.)"""
    if _get_raw_stream is not None:
        return _get_raw_stream(_get_device())
    return torch.cuda.current_stream().cuda_stream


def ptr(t):
    """Device pointer of a contiguous CUDA(HIP) tensor, or NULL for None."""
    if t is None:
        return None
    if not t.is_cuda:
        raise FscError("libfsc_hip kernels need device tensors; got a %s tensor. The accelerated "
                       "path has no CPU fallback." % t.device)
    if not t.is_contiguous():
        raise FscError("non-contiguous tensor passed to a libfsc_hip kernel")
    if t.device.index is not None and t.device.index != _get_device():
        # kernels are launched on the CURRENT device's stream (stream_ptr): a tensor of another GPU would be read through
        # peer access or fault.  Callers select the device first (torch.cuda.set_device / `with torch.cuda.device(...)`).
        raise FscError("tensor on %s passed to a libfsc_hip kernel while the current device is cuda:%d"
                       % (t.device, torch.cuda.current_device()))
    return t.data_ptr()


CALLS = [0]          # entry-point calls so far (bench.py reports calls per step: the launch-bound workloads are priced by it)
ON_ERROR = []        # callables run when an entry point fails (functional.py drops its pooled BatchNorm workspaces: an aborted call
                     # may have left their arrival counters non-zero)


def call(name, *args):
    lib = load()
    CALLS[0] += 1
    rc = getattr(lib, name)(*args)
    if rc != 0:
        msg = lib.fsc_last_error_string().decode()
        for hook in ON_ERROR:
            hook()
        raise FscError("%s failed (%d): %s" % (name, rc, msg))
