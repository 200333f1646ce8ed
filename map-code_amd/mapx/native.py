"""ctypes binding of libmapx_hip.so (C ABI: include/mapx_hip.h).

The library is the product: there is no Python/CPU fallback.  Importing this module
without the built .so raises; calling a kernel without a GPU raises.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmapx_hip.so")

MAPX_ABI_VERSION = 46
EPI_NONE, EPI_BIAS, EPI_BIAS_RELU, EPI_BIAS_CROSS, EPI_ADD, EPI_RELU_MASK, EPI_RELU_MASK_COLSUM = range(7)

_p, _i, _i64, _u64, _f, _d, _sz = (C.c_void_p, C.c_int, C.c_int64, C.c_uint64, C.c_float, C.c_double,
                                   C.c_size_t)

# name -> (restype, argtypes); mirrors include/mapx_hip.h one-to-one
SIGNATURES = {
    "mapx_last_error": (C.c_char_p, []),
    "mapx_abi_version": (_i, []),
    "mapx_emb_gather_fwd": (_i, [_p, _i64, _p, _i64, _i, _p, _p, _p, _p, _p]),
    "mapx_ids_to_i32": (_i, [_p, _i64, _i64, _p, _p, _p]),
    "mapx_seg_plan_workspace_bytes": (_sz, [_i64, _i64]),
    "mapx_seg_plan": (_i, [_p, _i64, _i64, _p, _sz, _p, _p, _p, _p, _p, _p, _p]),
    "mapx_seg_plan_multi_workspace_bytes": (_sz, [_i, _p, _p]),
    "mapx_seg_plan_multi": (_i, [_i, _p, _p, _p, _p, _sz, _p, _p, _p, _p, _p, _p, _p]),
    "mapx_seg_plan_merge": (_i, [_p, _i, _i64, _p, _sz, _p, _p, _p, _p, _p, _p, _p]),
    "mapx_seg_reduce_workspace_bytes": (_sz, [_i64, _i]),
    "mapx_seg_reduce_rows": (_i, [_i64, _p, _p, _p, _p, _p, _i, _p, _p, _sz, _p, _p]),
    "mapx_seg_reduce_rows_extra": (_i, [_i64, _p, _p, _p, _p, _i, _i64, _p, _i, _i64, _p, _p, _p, _sz, _p, _p]),
    "mapx_pack_sparse": (_i, [_p, _p, _i, _p, _p, _i64, _i64, _f, C.c_int32, _p, _p, _p]),
    "mapx_publish_i32": (_i, [_p, _i, _p, _p, _p]),
    "mapx_transpose_batched": (_i, [_p, _i64, _i, _i, _p, _p]),
    "mapx_cin_outer_fwd": (_i, [_p, _i, _p, _i, _i64, _p, _i64, _p]),
    "mapx_cin_outer_bwd": (_i, [_p, _i64, _p, _i, _p, _i, _i64, _p, _i, _p, _p]),
    "mapx_cin_pool_fwd": (_i, [_p, _i64, _i, _i, _p, _i64, _p]),
    "mapx_cin_pool_bwd": (_i, [_p, _i64, _i64, _i, _i, _p, _i, _p]),
    "mapx_host_alloc_coherent": (_i, [C.c_size_t, C.POINTER(C.c_void_p)]),
    "mapx_host_free": (_i, [_p]),
    "mapx_lr_sum_fwd": (_i, [_p, _i64, _i, _p, _i64, _p, _p, _p]),
    "mapx_fm_fwd": (_i, [_p, _i64, _i, _i, _p, _p, _p]),
    "mapx_fm_bwd": (_i, [_p, _p, _p, _i64, _i, _i, _p, _p]),
    "mapx_attn_fwd": (_i, [_p, _p, _p, _i64, _i, _i, _i, _p, _p, _p]),
    "mapx_attn_bwd": (_i, [_p, _p, _p, _p, _p, _i64, _i, _i, _i, _p, _p, _p, _p]),
    "mapx_alias_build_host": (_i, [_p, _i64, _p, _p]),
    "mapx_alias_pack": (_i, [_p, _p, _i64, _p, _p]),
    "mapx_alias_draw": (_i, [_p, _i64, _p, _i64, _i, _u64, _u64, _p, _p, _p]),
    "mapx_nce_pack_idx": (_i, [_p, _p, _i64, _i, _i64, _p, _p, _p]),
    "mapx_nce_fwd_workspace_bytes": (_sz, []),
    "mapx_nce_fwd": (_i, [_p, _i64, _i, _i, _i, _p, _p, _i, _p, _p, _p, _i64, _p, _p, _p, _p, _p,
                          _p, _p, _sz, _p, _p, _p, _p, _p, _p]),
    "mapx_nce_scatter_dh": (_i, [_p, _p, _p, _i64, _i, _i, _i, _p, _p, _i, _p, _p, _p, _p]),
    "mapx_nce_table_grad_workspace_bytes": (_sz, [_i64, _i]),
    "mapx_nce_table_grad": (_i, [_i64, _p, _p, _p, _p, _p, _i, _i, _p, _p, _p, _p, _sz, _p, _p]),
    "mapx_scale_inplace": (_i, [_p, _i64, _p, _p]),
    "mapx_gemm_splitk_workspace_bytes": (_sz, [_i, _i, _i]),
    "mapx_gemm_f32": (_i, [_i, _i, _i, _i, _i, _p, _i64, _p, _i64, _p, _i64, _i, _p, _p, _i64, _p,
                           _i64, _p, _i64, _i, _i, _p, _sz, _p, _p, _p]),
    "mapx_amax_epoch_source": (_i, [_p]),
    "mapx_h2_weight_planes_bytes": (_sz, [_i, _i]),
    "mapx_h2_weight_planes": (_i, [_p, _i64, _i, _i, _i, _p, _p, _p]),
    "mapx_h2_weight_planes_multi": (_i, [_p, _i, _p]),
    "mapx_amax_f32": (_i, [_p, _i64, _i64, _i64, _p, _i, _p]),
    "mapx_gemm_f32_bwd_fused": (_i, [_i, _i, _i, _p, _i64, _p, _i64, _p, _i64, _p, _i64, _p, _i64, _i, _p, _i64, _p, _i64,
                                     _p, _i64, _p, _i64, _i, _i, _p, _i64, _p, _p]),
    "mapx_gemm_f32_batched": (_i, [_i, _i, _i, _i, _i, _i, _p, _i64, _p, _i64, _p, _i, _p, _sz, _p, _p]),
    "mapx_gemm_bf16": (_i, [_i, _i, _i, _i, _i, _p, _i64, _p, _i64, _p, _i64, _i, _i, _p, _p, _i64, _i, _p, _i64,
                            _p, _i64, _i, _i, _p, _sz, _p]),
    "mapx_cast_f32_bf16": (_i, [_p, _i64, _p, _p]),
    "mapx_cast_bf16_f32": (_i, [_p, _i64, _p, _p]),
    "mapx_emb_gather_fwd_bf16": (_i, [_p, _i64, _p, _i64, _i, _p, _p, _p, _p]),
    "mapx_seg_reduce_rows_bf16": (_i, [_i64, _p, _p, _p, _p, _p, _i, _p, _p, _sz, _p, _p]),
    "mapx_colsum_bf16_workspace_bytes": (_sz, [_i]),
    "mapx_colsum_bf16": (_i, [_p, _i64, _i, _i, _p, _p, _sz, _p]),
    "mapx_relu_mask_colsum_bf16": (_i, [_p, _i64, _p, _i64, _i, _i, _p, _p, _p, _sz, _p]),
    "mapx_cross_bwd_pre_colsum_bf16": (_i, [_p, _i64, _p, _p, _i, _i, _p, _p, _i, _p, _p, _sz, _p]),
    "mapx_relu_mask_bf16": (_i, [_p, _p, _i64, _p, _p]),
    "mapx_adamw_dense_shadow": (_i, [_p, _p, _p, _p, _i64, _p, _i, _p, _d, _d, _d, _d, _p, _p]),
    "mapx_dropout": (_i, [_p, _i64, _f, _u64, _u64, _p, _p, _p]),
    "mapx_layernorm_fwd": (_i, [_p, _i64, _i, _p, _p, _f, _p, _p, _p]),
    "mapx_layernorm_bwd": (_i, [_p, _p, _p, _p, _i64, _i, _p, _p, _p]),
    "mapx_sum_tasks": (_i, [_p, _i, _p]),
    "mapx_skinny_chunks": (_i, []),
    "mapx_skinny_linear_fwd": (_i, [_p, _i64, _p, _i64, _p, _i, _i, _i, _i, _p, _i64, _p]),
    "mapx_skinny_linear_dw": (_i, [_p, _i64, _p, _i64, _i, _i, _i, _p, _i, _p]),
    "mapx_skinny_linear_dx": (_i, [_p, _i64, _p, _i64, _i, _i, _i, _p, _i64, _p]),
    "mapx_skinny_linear_fwd_bf16": (_i, [_p, _i64, _p, _i64, _p, _i, _i, _i, _i, _p, _i64, _p]),
    "mapx_skinny_linear_dw_bf16": (_i, [_p, _i64, _p, _i64, _i, _i, _i, _p, _i, _p]),
    "mapx_skinny_linear_dx_bf16": (_i, [_p, _i64, _p, _i64, _i, _i, _i, _p, _i64, _p]),
    "mapx_skinny_join_bwd": (_i, [_p, _i64, _p, _i64, _i, _i, _i, _i, _p, _i64, _p, _i64, _p, _i64, _i, _p, _i64, _p, _i64,
                                  _p, _i64, _p, _i64, _p, _p, _p, _p, _p]),
    "mapx_enc_group_layout": (_i, [_p, _i, _i, _i, _i, _p, _p, _p, _p, _p]),
    "mapx_enc_grouped_fwd": (_i, [_p, _i64, _i, _i, _p, _i64, _p, _p, _p, _p, _i, _i, _p, _p, _p, _p]),
    "mapx_enc_grouped_dw": (_i, [_p, _p, _i64, _i, _i, _p, _p, _i, _p, _p, _i64, _p, _p]),
    "mapx_colsum_chunks": (_i, []),
    "mapx_colsum_workspace_bytes": (_sz, [_i]),
    "mapx_colsum": (_i, [_p, _i64, _i, _i, _p, _p, _sz, _p]),
    "mapx_cross_bwd_pre": (_i, [_p, _p, _p, _i64, _p, _p, _i, _p]),
    "mapx_relu_mask": (_i, [_p, _p, _i64, _p, _p]),
    "mapx_relu_mask_colsum": (_i, [_p, _i64, _p, _i64, _i, _i, _p, _p, _p, _sz, _p, _p]),
    "mapx_cross_bwd_pre_colsum": (_i, [_p, _i64, _p, _p, _i, _i, _p, _p, _i, _p, _p, _sz, _p, _p]),
    "mapx_bce_workspace_bytes": (_sz, []),
    "mapx_bce_with_logits": (_i, [_p, _p, _i64, _p, _p, _p, _sz, _p]),
    "mapx_eval_metrics_workspace_bytes": (_sz, [_i64]),
    "mapx_eval_metrics": (_i, [_p, _p, _i64, _p, _p, _sz, _p]),
    "mapx_dynamic_mask_mfp": (_i, [_p, _i64, _i, _i, _p, _u64, _u64, _p, _p, _p, _p, _p, _p]),
    "mapx_dynamic_mask_mfp_rows": (_i, [_p, _i64, _p, _i64, _p, _i64, _i, _i, _p, _u64, _u64, _p, _p, _p, _p, _p, _p]),
    "mapx_dynamic_mask_rfd": (_i, [_p, _i64, _i, _i, _p, _p, _p, _i64, _i, _p, _p, _i64, _u64, _u64, _p, _p, _p,
                                  _p, _p]),
    "mapx_adamw_dense": (_i, [_p, _p, _p, _p, _i64, _p, _i, _p, _d, _d, _d, _d, _p, _i, _p, _p]),
    "mapx_step_advance": (_i, [_p, _p, _i64, _p]),
    "mapx_take_rows_i64": (_i, [_p, _i64, _i, _p, _i64, _p, _i64, _p, _p, _p]),
    "mapx_act_fwd": (_i, [_i, _p, _i64, _i, _p, _i64, _p]),
    "mapx_act_bwd": (_i, [_i, _p, _i64, _p, _i64, _i, _p, _p]),
    "mapx_vocab_table_init": (_i, [_p, _p, _p, _i64, _p]),
    "mapx_vocab_count": (_i, [_p, _i64, _p, _p, _p, _i64, _p, _p, _p]),
    "mapx_vocab_compact": (_i, [_p, _p, _p, _i64, _p, _p, _p, _p, _p, _p]),
    "mapx_vocab_rank_keys": (_i, [_p, _p, _i64, C.c_int32, _p, _p]),
    "mapx_vocab_assign": (_i, [_p, _p, _p, _p, _i64, C.c_int32, _p, _p, _p, _p, _p]),
    "mapx_vocab_map": (_i, [_p, _p, _p, _p, _i64, _i64, _p, _i64, _p]),
    "mapx_replay_coef_table_bytes": (_sz, [_i]),
    "mapx_replay_coef_table": (_i, [_p, _i, _i, _d, _d, _p, _p, _p]),
    "mapx_table_adam": (_i, [_p, _p, _p, _i64, _i, _f, _p, _p, _p, _i64, _f, _p, _p, _i64, _i64, _p, _p, _p, _p,
                             _i, _p, _p, _i, _i, _d, _d, _d, _i, _p]),
}


class SumTask(C.Structure):
    """mapx_sum_task (include/mapx_hip.h)."""
    _fields_ = [("dst", _p), ("src", _p), ("stride", _i64), ("n", _i64), ("nsplit", C.c_int32),
                ("pad_", C.c_int32)]


class PlaneTask(C.Structure):
    """mapx_plane_task (include/mapx_hip.h)."""
    _fields_ = [("W", _p), ("ldw", _i64), ("N", C.c_int32), ("K", C.c_int32), ("b_kc", C.c_int32), ("pad_", C.c_int32),
                ("amax_record", _p), ("planes", _p)]


class GemmScale(C.Structure):
    """mapx_gemm_scale (include/mapx_hip.h): magnitude records of a product's operands and outputs."""
    _fields_ = [("amax_a", _p), ("amax_b", _p), ("amax_c", _p), ("amax_c2", _p), ("b_planes", _p)]


class LazyRows(C.Structure):
    """mapx_lazy_rows (include/mapx_hip.h): the lazy-AdamW state of the table rows a forward kernel reads through
    their pending zero-gradient updates."""
    _fields_ = [("m0", _p), ("v0", _p), ("ld_mv0", _i64), ("wd0", _f),
                ("m1", _p), ("v1", _p), ("ld_mv1", _i64), ("wd1", _f),
                ("last", _p), ("sched", _p), ("sched_len", _i), ("done", _p),
                ("aux", _p), ("aux_len", _i), ("aux_rows", _i),
                ("beta1", _d), ("beta2", _d), ("eps", _d), ("coef_opt", _p)]


class MapxError(RuntimeError):
    pass


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `make -C map-code_amd/csrc` "
            "(or __graft_entry__.build()).  There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is not exported
        fn.restype, fn.argtypes = res, args
    got = lib.mapx_abi_version()
    if got != MAPX_ABI_VERSION:
        raise ImportError(f"libmapx_hip.so ABI {got} != binding ABI {MAPX_ABI_VERSION}")
    return lib


lib = _load()


def check(status):
    if status != 0:
        raise MapxError(f"mapx status {status}: {lib.mapx_last_error().decode()}")


def require_gpu(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise MapxError("mapx kernels need device (HIP) tensors; there is no CPU path")


def ptr(t):
    """Device pointer of a contiguous tensor (None -> NULL)."""
    if t is None:
        return None
    assert t.is_contiguous(), "mapx kernels take contiguous tensors"
    return t.data_ptr()


def stream():
    return torch.cuda.current_stream().cuda_stream
