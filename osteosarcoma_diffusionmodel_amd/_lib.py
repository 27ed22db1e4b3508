"""ctypes binding of libosdiff.so (C ABI: include/osdiff.h).

There is deliberately no fallback: if the shared library is missing, or a call
fails, an exception is raised -- the HIP path is the product.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

OSD_MAX_HIDDEN = 8
OSD_F_GRAPH, OSD_F_TRAIN_MODE, OSD_F_SYNC = 1, 2, 4
OSD_OK, OSD_EINVAL, OSD_ENOMEM, OSD_EHIP, OSD_ESTATE, OSD_EUNSUPPORTED = 0, -1, -2, -3, -4, -5
OSD_COMM_ID_BYTES = 128

LIB_PATH = Path(__file__).resolve().parent / "lib" / "libosdiff.so"


class OsdConfig(C.Structure):
    _fields_ = [
        ("mutation_dim", C.c_int32), ("expression_dim", C.c_int32), ("pathway_dim", C.c_int32),
        ("condition_dim", C.c_int32), ("time_dim", C.c_int32), ("n_hidden", C.c_int32),
        ("hidden_dims", C.c_int32 * OSD_MAX_HIDDEN), ("num_steps", C.c_int32),
        ("dropout_p", C.c_float), ("device", C.c_int32),
    ]


class OsdConstraints(C.Structure):
    _fields_ = [
        ("pathway_offsets", C.POINTER(C.c_int32)), ("pathway_members", C.POINTER(C.c_int32)), ("n_pathways", C.c_int32),
        ("pathway_weight", C.c_double),
        ("cols_a", C.POINTER(C.c_int32)), ("cols_b", C.POINTER(C.c_int32)), ("n_a", C.c_int32), ("n_b", C.c_int32),
        ("mutexpr_weight", C.c_double),
    ]


_P = C.c_void_p
_SIGNATURES = {
    "osd_version": (C.c_int, []),
    "osd_last_error": (C.c_char_p, []),
    "osd_num_params": (C.c_int, [C.POINTER(OsdConfig)]),
    "osd_param_numel": (C.c_int64, [C.POINTER(OsdConfig), C.c_int]),
    "osd_create": (C.c_int, [C.POINTER(OsdConfig), C.POINTER(_P)]),
    "osd_destroy": (C.c_int, [_P]),
    "osd_set_stream": (C.c_int, [_P, _P]),
    "osd_set_option": (C.c_int, [_P, C.c_char_p, C.c_int64]),
    "osd_get_option": (C.c_int, [_P, C.c_char_p, C.POINTER(C.c_int64)]),
    "osd_set_schedule": (C.c_int, [_P, _P, _P, _P, _P]),
    "osd_load_weights": (C.c_int, [_P, C.POINTER(_P), C.c_int]),
    "osd_denoiser_forward": (C.c_int, [_P, _P, _P, C.c_int32, _P, C.c_int64, _P, C.c_int, C.POINTER(_P), C.c_uint64]),
    "osd_q_sample": (C.c_int, [_P, _P, _P, _P, C.c_int64, C.c_uint64, C.c_int64, _P, _P]),
    "osd_p_sample_step": (C.c_int, [_P, _P, C.c_int32, _P, _P, C.c_int64, C.c_uint64, C.c_int64, _P, C.c_int]),
    "osd_sample_chain": (C.c_int, [_P, _P, C.c_int64, _P, _P, C.c_uint64, C.c_int64, _P, _P, C.c_int]),
    "osd_sample_engine": (C.c_int, [_P, C.c_int64, C.c_int]),
    "osd_train_loss_fwd_bwd": (C.c_int, [_P, _P, _P, C.c_int64, _P, _P, C.POINTER(_P), C.c_uint64, C.c_int64, C.c_int,
                                         _P, C.POINTER(_P), C.c_double, C.POINTER(_P), C.c_int]),
    "osd_train_batch_source": (C.c_int, [_P, _P, C.c_int64, _P, C.c_int64, _P, _P, C.c_double]),
    "osd_denoiser_forward_train": (C.c_int, [_P, _P, _P, _P, C.c_int64, C.POINTER(_P), C.c_uint64, C.c_int64, C.c_int, _P]),
    "osd_denoiser_backward": (C.c_int, [_P, _P, _P, _P, C.c_int64, _P, C.POINTER(_P), C.c_uint64, C.c_int64, C.c_int, C.POINTER(_P), _P,
                                        C.POINTER(_P), C.c_int]),
    "osd_grad_buckets": (C.c_int, [C.POINTER(OsdConfig), C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.c_int]),
    "osd_comm_unique_id": (C.c_int, [_P]),
    "osd_comm_create": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.POINTER(_P)]),
    "osd_comm_destroy": (C.c_int, [_P]),
    "osd_allreduce_grads_begin": (C.c_int, [_P, _P, _P, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(_P), C.c_int]),
    "osd_allreduce_grads_end": (C.c_int, [_P, _P]),
    "osd_mixup": (C.c_int, [_P, _P, _P, _P, _P, C.c_double, C.c_int64, _P, _P, _P]),
    "osd_clip_adamw_step": (C.c_int, [_P, _P, _P, _P, _P, C.c_int64, C.c_double, C.c_double, C.c_double, C.c_double,
                                      C.c_double, C.c_double, C.c_int64, _P]),
    "osd_val_mmd": (C.c_int, [_P, C.c_int, _P, C.c_int64, _P, C.c_int64, C.c_int, C.c_double, C.POINTER(C.c_double)]),
    "osd_val_ks_extremes": (C.c_int, [_P, C.c_int, _P, C.c_int64, _P, C.c_int64, C.c_int, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "osd_val_mean_offdiag_corr": (C.c_int, [_P, C.c_int, _P, C.c_int64, C.c_int, C.POINTER(C.c_int32), C.c_int, C.POINTER(C.c_double)]),
    "osd_val_column_sums": (C.c_int, [_P, C.c_int, _P, C.c_int64, C.c_int, C.c_int, C.POINTER(C.c_double)]),
    "osd_val_gram": (C.c_int, [_P, C.c_int, _P, C.c_int64, C.c_int, C.POINTER(C.c_int32), C.c_int, C.POINTER(C.c_double)]),
    "osd_val_rbf_sum": (C.c_int, [_P, C.c_int, _P, C.c_int64, _P, C.c_int64, C.c_int, C.c_double, C.POINTER(C.c_double)]),
    "osd_val_col_moments": (C.c_int, [_P, C.c_int, _P, C.c_int64, C.c_int, C.POINTER(C.c_int32), C.c_int, C.POINTER(C.c_double),
                                      C.POINTER(C.c_double)]),
    "osd_val_rowz_sq": (C.c_int, [_P, C.c_int, _P, C.c_int64, C.c_int, C.POINTER(C.c_int32), C.c_int, C.POINTER(C.c_double),
                                  C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "osd_val_pearson_sums": (C.c_int, [_P, C.c_int, _P, C.c_int, _P, C.c_int, C.c_int64, C.POINTER(C.c_double)]),
    "osd_val_pearson": (C.c_int, [_P, C.c_int, _P, C.c_int, _P, C.c_int, C.c_int64, C.POINTER(C.c_double)]),
    "osd_set_constraints": (C.c_int, [_P, C.POINTER(OsdConstraints)]),
    "osd_get_loss_parts": (C.c_int, [_P, C.POINTER(C.c_float)]),
    "osd_loss_pathway_coherence": (C.c_int, [_P, C.c_int, _P, C.c_int64, C.c_int, C.c_int, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.c_int,
                                             C.c_double, _P, _P]),
    "osd_loss_mutation_expression": (C.c_int, [_P, C.c_int, _P, _P, C.c_int64, C.c_int, C.c_int, C.POINTER(C.c_int32), C.c_int,
                                               C.POINTER(C.c_int32), C.c_int, C.c_double, _P, _P]),
    "osd_nn_linear": (C.c_int, [_P, C.c_int, _P, C.c_int, _P, C.c_int, _P, _P, C.c_int64, C.c_int, _P]),
    "osd_nn_linear_bwd": (C.c_int, [_P, C.c_int, _P, C.c_int, _P, C.c_int, _P, _P, C.c_int64, C.c_int, _P, _P, _P]),
    "osd_nn_bn_relu_dropout": (C.c_int, [_P, C.c_int, _P, C.c_int64, C.c_int, _P, _P, _P, _P, C.c_double, C.c_double, C.c_int, C.c_int,
                                         C.c_double, _P, C.c_uint64, C.c_uint32, _P, _P, _P]),
    "osd_nn_bn_relu_dropout_bwd": (C.c_int, [_P, C.c_int, _P, _P, C.c_int64, C.c_int, _P, _P, _P, _P, C.c_int, C.c_int, C.c_double, _P,
                                             C.c_uint64, C.c_uint32, _P, _P, _P]),
    "osd_nn_reparameterize": (C.c_int, [_P, C.c_int, _P, _P, _P, C.c_uint64, C.c_int64, C.c_int, _P, _P]),
    "osd_nn_reparameterize_bwd": (C.c_int, [_P, C.c_int, _P, _P, _P, C.c_int64, _P]),
    "osd_nn_clip_adamw_step": (C.c_int, [_P, C.c_int, _P, _P, _P, _P, _P, C.c_int64, C.c_double, C.c_double, C.c_double, C.c_double,
                                         C.c_double, C.c_double, C.c_int64, _P]),
    "osd_nn_vae_loss": (C.c_int, [_P, C.c_int, _P, _P, _P, _P, C.c_int64, C.c_int, C.c_int, _P, _P, _P, _P]),
    "osd_nn_mixup": (C.c_int, [_P, C.c_int, _P, _P, C.c_double, C.c_int64, C.c_int, _P]),
    "osd_nn_mixup3": (C.c_int, [_P, C.c_int, _P, _P, _P, _P, C.c_double, C.c_int64, C.c_int, C.c_int, _P, _P, _P]),
    "osd_nn_mse": (C.c_int, [_P, C.c_int, _P, _P, C.c_int64, _P, _P]),
    "osd_profile_step": (C.c_int, [_P, _P, C.c_int64, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_double), C.c_int, C.POINTER(C.c_int)]),
    "osd_op_linear": (C.c_int, [_P, _P, _P, _P, C.c_int64, C.c_int, C.c_int, C.c_int, _P]),
    "osd_op_linear_gn_silu": (C.c_int, [_P, _P, C.c_int, _P, C.c_int, _P, _P, _P, _P, C.c_int64, C.c_int, _P]),
    "osd_op_gemm": (C.c_int, [_P, _P, C.c_int, C.c_int, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P, C.c_int, C.c_int]),
    "osd_op_randn": (C.c_int, [_P, _P, C.c_int64, C.c_int, C.c_uint64, C.c_int64, C.c_uint32, C.c_uint32]),
}

_lib = None


def lib() -> C.CDLL:
    """Load libosdiff.so (once).  Raises if it has not been built."""
    global _lib
    if _lib is None:
        path = Path(os.environ.get("OSDIFF_LIB", LIB_PATH))
        if not path.exists():
            raise RuntimeError(
                f"{path} not found: build the HIP library first "
                "(make -C osteosarcoma_diffusionmodel_amd/csrc, or __graft_entry__.build()); "
                "this package has no CPU fallback")
        handle = C.CDLL(str(path))
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(handle, name)      # AttributeError if the ABI lost a symbol
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


def exported_symbols():
    return list(_SIGNATURES)


def last_error() -> str:
    msg = lib().osd_last_error()
    return msg.decode("utf-8", "replace") if msg else ""


def check(rc: int) -> None:
    """Map a negative OSD_E* return code to the exception the reference would raise."""
    if rc == OSD_OK:
        return
    msg = last_error() or f"libosdiff error {rc}"
    if rc in (OSD_EINVAL, OSD_EUNSUPPORTED):
        raise ValueError(msg)
    if rc == OSD_ENOMEM:
        raise MemoryError(msg)
    raise RuntimeError(msg)


def ptr(t) -> C.c_void_p:
    """Device (or host) address of a tensor, or NULL for None."""
    return C.c_void_p(0 if t is None else t.data_ptr())


def ptr_array(tensors):
    arr = (C.c_void_p * len(tensors))()
    for i, t in enumerate(tensors):
        arr[i] = 0 if t is None else t.data_ptr()
    return arr
