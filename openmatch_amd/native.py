"""ctypes binding of libopenmatch_hip.so (C ABI declared in include/openmatch_hip.h).

The product path has NO fallback: if the shared library is missing, was built for another
ABI version, or a tensor does not live on an MI355X device, the call raises.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libopenmatch_hip.so")

OM_F32, OM_BF16, OM_F16 = 0, 1, 2
ACT_NONE, ACT_GELU_ERF, ACT_RELU, ACT_GELU_TANH = 0, 1, 2, 3
ACT_MUL_RESID = 0x100
ARCH_BERT, ARCH_T5 = 0, 1
POOL_NONE, POOL_FIRST, POOL_MEAN = 0, 1, 2
# om_debug_option switches used from Python (include/openmatch_hip.h: OM_OPT_*)
OPT_TRAIN_WGRAD_BATCH, OPT_GEMM_MAX_GRID, OPT_GEMM_CONT = 14, 15, 16
SEARCH_F32, SEARCH_F16_RESCORE = 0, 1
ABI_VERSION = 5

c_void_p, c_int, c_int64, c_float, c_size_t = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_size_t


class OmLayerWeights(C.Structure):
    _fields_ = [(n, c_void_p) for n in (
        "qkv_w", "qkv_b", "o_w", "o_b", "ln1_g", "ln1_b", "ffn1_w", "ffn1_b", "ffn1g_w",
        "ffn2_w", "ffn2_b", "ln2_g", "ln2_b")]


class OmEncoderConfig(C.Structure):
    _fields_ = [("arch", c_int), ("dtype", c_int), ("hidden", c_int), ("n_layers", c_int),
                ("n_heads", c_int), ("head_dim", c_int), ("ffn", c_int), ("vocab", c_int),
                ("max_pos", c_int), ("type_vocab", c_int), ("act", c_int), ("ln_eps", c_float),
                ("rel_buckets", c_int), ("rel_max_dist", c_int), ("pooling", c_int),
                ("head_in", c_int), ("head_out", c_int), ("normalize", c_int)]


class OmEncoderWeights(C.Structure):
    _fields_ = [("word_emb", c_void_p), ("pos_emb", c_void_p), ("type_emb", c_void_p),
                ("emb_ln_g", c_void_p), ("emb_ln_b", c_void_p),
                ("layers_host", C.POINTER(OmLayerWeights)), ("final_ln_g", c_void_p),
                ("rel_bias", c_void_p), ("head_w", c_void_p), ("folded", c_void_p)]


class OmTnProblem(C.Structure):
    """One weight-gradient contraction of an om_gemm_tn_acc_batch launch (include/openmatch_hip.h)."""
    _fields_ = [("A", c_void_p), ("B", c_void_p), ("C", c_void_p), ("bias", c_void_p),
                ("lda", c_int64), ("ldb", c_int64), ("ldc", c_int64), ("N", c_int64), ("K", c_int64)]


class OmT5DecoderLayer(C.Structure):
    _fields_ = [(n, c_void_p) for n in (
        "sa_v_w", "sa_o_w", "sa_ln_g", "ca_q_w", "ca_kv_w", "ca_o_w", "ca_ln_g", "ffn1_w", "ffn1g_w", "ffn2_w", "ffn_ln_g")]


class OmT5DecoderWeights(C.Structure):
    _fields_ = [("start_emb", c_void_p), ("final_ln_g", c_void_p),
                ("layers_host", C.POINTER(OmT5DecoderLayer)), ("n_layers", c_int)]


class OmT5DecoderLayerGrads(C.Structure):
    _fields_ = [(n, c_void_p) for n in (
        "sa_v_w", "sa_o_w", "sa_ln_g", "ca_q_w", "ca_kv_w", "ca_o_w", "ca_ln_g", "ffn1_w", "ffn1g_w", "ffn2_w", "ffn_ln_g")]


class OmT5DecoderGrads(C.Structure):
    _fields_ = [("start_emb", c_void_p), ("final_ln_g", c_void_p), ("layers_host", C.POINTER(OmT5DecoderLayerGrads))]


class OmAdamTensor(C.Structure):
    """One parameter of an om_adamw_step launch (include/openmatch_hip.h)."""
    _fields_ = [("p", c_void_p), ("g", c_void_p), ("m", c_void_p), ("v", c_void_p), ("shadow0", c_void_p), ("shadow1", c_void_p),
                ("n", c_int64), ("weight_decay", c_float), ("shadow0_dtype", c_int), ("shadow1_dtype", c_int), ("reserved", c_int)]


ADAM_CHUNK = 16384


class OmLayerGrads(C.Structure):
    _fields_ = [(n, c_void_p) for n in (
        "qkv_w", "qkv_b", "o_w", "o_b", "ln1_g", "ln1_b", "ffn1_w", "ffn1_b", "ffn2_w", "ffn2_b",
        "ln2_g", "ln2_b", "ffn1g_w")]


class OmEncoderGrads(C.Structure):
    _fields_ = [("word_emb", c_void_p), ("pos_emb", c_void_p), ("type_emb", c_void_p),
                ("emb_ln_g", c_void_p), ("emb_ln_b", c_void_p),
                ("layers_host", C.POINTER(OmLayerGrads)), ("head_w", c_void_p),
                ("final_ln_g", c_void_p), ("rel_bias", c_void_p)]


_SIGNATURES = {
    "om_last_error": (C.c_char_p, []),
    "om_abi_version": (c_int, []),
    "om_device_count": (c_int, []),
    "om_debug_gemm_trace": (None, [c_void_p]),
    "om_debug_gemm_gen": (None, [c_int]),
    "om_debug_option": (c_int, [c_int, c_int]),
    "om_debug_attention": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_void_p]),
    "om_debug_wave_sum_check": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "om_encoder_fold_bytes": (c_size_t, [C.POINTER(OmEncoderConfig)]),
    "om_encoder_fold_weights": (c_int, [C.POINTER(OmEncoderConfig), C.POINTER(OmEncoderWeights), c_void_p, c_size_t, c_void_p]),
    "om_kernel_timing_enable": (c_int, [c_int]),
    "om_kernel_timing_read": (c_int, [c_int, C.POINTER(C.c_double), C.POINTER(c_int64), C.POINTER(C.c_double)]),
    "om_gemm_nt": (c_int, [c_int, c_void_p, c_int64, c_void_p, c_int64, c_int, c_void_p, c_int64,
                           c_int64, c_int64, c_int64, c_void_p, c_void_p, c_int64, c_int, c_void_p]),
    "om_gemm_tn_acc": (c_int, [c_int, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_void_p,
                               c_int64, c_int64, c_int64, c_void_p]),
    "om_gemm_tn_acc_batch": (c_int, [c_int, C.POINTER(OmTnProblem), c_int, c_int64, c_void_p]),
    "om_encoder_workspace_bytes": (c_size_t, [C.POINTER(OmEncoderConfig), c_int64, c_int64]),
    "om_t5_decoder_workspace_bytes": (c_size_t, [C.POINTER(OmEncoderConfig), c_int64, c_int64]),
    "om_t5_decoder_step": (c_int, [C.POINTER(OmEncoderConfig), C.POINTER(OmT5DecoderWeights), c_void_p, c_void_p,
                                   c_int64, c_int64, c_void_p, c_void_p, c_size_t, c_void_p]),
    "om_t5_relative_bucket": (c_int, [c_int, c_int, c_int]),
    "om_encoder_forward": (c_int, [C.POINTER(OmEncoderConfig), C.POINTER(OmEncoderWeights), c_void_p,
                                   c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_void_p,
                                   c_void_p, c_size_t, c_void_p]),
    "om_encoder_packed_supported": (c_int, [C.POINTER(OmEncoderConfig), c_int, c_int64, c_int64, c_int64]),
    "om_encoder_workspace_bytes_packed": (c_size_t, [C.POINTER(OmEncoderConfig), c_int64, c_int64, c_int64]),
    "om_encoder_forward_packed": (c_int, [C.POINTER(OmEncoderConfig), C.POINTER(OmEncoderWeights), c_void_p,
                                          c_void_p, c_void_p, c_int64, c_int64, c_int64, c_void_p,
                                          c_void_p, c_size_t, c_void_p]),
    "om_encoder_tape_bytes": (c_size_t, [C.POINTER(OmEncoderConfig), c_int64, c_int64]),
    "om_encoder_train_workspace_bytes": (c_size_t, [C.POINTER(OmEncoderConfig), c_int64, c_int64]),
    "om_encoder_train_forward": (c_int, [C.POINTER(OmEncoderConfig), C.POINTER(OmEncoderWeights), c_void_p,
                                         c_void_p, c_void_p, c_int64, c_int64, c_float, c_float, C.c_uint64,
                                         c_void_p, c_size_t, c_void_p, c_void_p, c_size_t, c_void_p]),
    "om_encoder_train_backward": (c_int, [C.POINTER(OmEncoderConfig), C.POINTER(OmEncoderWeights), c_void_p,
                                          c_void_p, c_void_p, c_int64, c_int64, c_float, c_float, C.c_uint64,
                                          c_void_p, c_void_p, C.POINTER(OmEncoderGrads), c_void_p, c_size_t,
                                          c_void_p]),
    "om_encoder_train_packed_supported": (c_int, [C.POINTER(OmEncoderConfig), c_int64, c_int64, c_int64]),
    "om_encoder_tape_bytes_packed": (c_size_t, [C.POINTER(OmEncoderConfig), c_int64, c_int64, c_int64]),
    "om_encoder_train_workspace_bytes_packed": (c_size_t, [C.POINTER(OmEncoderConfig), c_int64, c_int64, c_int64]),
    "om_encoder_train_forward_packed": (c_int, [C.POINTER(OmEncoderConfig), C.POINTER(OmEncoderWeights), c_void_p,
                                                c_void_p, c_void_p, c_int64, c_int64, c_int64, c_float, c_float, C.c_uint64,
                                                c_void_p, c_size_t, c_void_p, c_void_p, c_size_t, c_void_p]),
    "om_encoder_train_backward_packed": (c_int, [C.POINTER(OmEncoderConfig), C.POINTER(OmEncoderWeights), c_void_p,
                                                 c_void_p, c_void_p, c_int64, c_int64, c_int64, c_float, c_float, C.c_uint64,
                                                 c_void_p, c_void_p, C.POINTER(OmEncoderGrads), c_void_p, c_size_t,
                                                 c_void_p]),
    "om_encoder_train_forward_hidden": (c_int, [C.POINTER(OmEncoderConfig), C.POINTER(OmEncoderWeights), c_void_p,
                                                c_void_p, c_void_p, c_int64, c_int64, c_float, c_float, C.c_uint64,
                                                c_void_p, c_size_t, c_void_p, c_void_p, c_size_t, c_void_p]),
    "om_encoder_train_backward_hidden": (c_int, [C.POINTER(OmEncoderConfig), C.POINTER(OmEncoderWeights), c_void_p,
                                                 c_void_p, c_void_p, c_int64, c_int64, c_float, c_float, C.c_uint64,
                                                 c_void_p, c_void_p, C.POINTER(OmEncoderGrads), c_void_p, c_size_t,
                                                 c_void_p]),
    "om_t5_decoder_tape_bytes": (c_size_t, [C.POINTER(OmEncoderConfig), c_int, c_int64, c_int64]),
    "om_t5_decoder_train_workspace_bytes": (c_size_t, [C.POINTER(OmEncoderConfig), c_int, c_int64, c_int64]),
    "om_t5_decoder_train_forward": (c_int, [C.POINTER(OmEncoderConfig), C.POINTER(OmT5DecoderWeights), c_void_p, c_void_p,
                                            c_int64, c_int64, c_float, C.c_uint64, c_void_p, c_size_t, c_void_p,
                                            c_void_p, c_size_t, c_void_p]),
    "om_t5_decoder_train_backward": (c_int, [C.POINTER(OmEncoderConfig), C.POINTER(OmT5DecoderWeights), c_void_p, c_void_p,
                                             c_int64, c_int64, c_float, C.c_uint64, c_void_p, c_void_p,
                                             C.POINTER(OmT5DecoderGrads), c_void_p, c_void_p, c_size_t, c_void_p]),
    "om_linear_f32_backward": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "om_index_to_f16": (c_int, [c_void_p, c_int64, c_int, c_void_p, c_void_p, c_void_p]),
    "om_sim_topk_workspace_bytes": (c_size_t, [c_int64, c_int, c_int]),
    "om_sim_topk": (c_int, [c_int, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int64, c_int,
                            c_int, c_int64, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "om_sim_topk_info": (None, [C.POINTER(c_int64)]),
    "om_topk_merge": (c_int, [c_void_p, c_void_p, c_int, c_int64, c_int, c_int, c_void_p, c_void_p,
                              c_void_p]),
    "om_contrastive_fwd_bwd": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float,
                                       c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p,
                                       c_void_p, c_void_p, c_void_p]),
    "om_encoder_train_set_layer_events": (c_int, [c_void_p, c_int]),
    "om_comm_unique_id": (c_int, [c_void_p]),
    "om_comm_init": (c_int, [c_void_p, c_int, c_int, C.POINTER(c_void_p)]),
    "om_comm_destroy": (c_int, [c_void_p]),
    "om_comm_count": (c_int, [c_void_p, C.POINTER(c_int)]),
    "om_allgather_rows": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_void_p]),
    "om_allreduce_grads": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_void_p]),
    "om_exchange_topk": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int64, c_int, c_void_p, c_void_p, c_void_p]),
    "om_grad_sqnorm": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
    "om_adamw_step": (c_int, [c_void_p, c_void_p, c_int, c_float, c_float, c_float, c_float, c_int64, c_void_p, c_float, c_float,
                              c_int, c_void_p, c_void_p]),
    "om_loss_scale_update": (c_int, [c_void_p, c_void_p, c_int, c_void_p]),
    "om_contrastive_fwd_bwd_ex": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_int, c_void_p,
                                          c_float, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p,
                                          c_void_p, c_void_p, c_void_p]),
}

_lib = None


class NativeError(RuntimeError):
    pass


def lib():
    """The loaded library (loads on first use; raises if it is not there)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise NativeError(
                f"{LIB_PATH} is missing: build the HIP library first "
                "(python -m openmatch_amd._build, or __graft_entry__.build()). "
                "openmatch_amd has no CPU / eager fallback.")
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            try:
                fn = getattr(handle, name)
            except AttributeError as e:
                raise NativeError(f"{LIB_PATH} does not export {name}; rebuild it") from e
            fn.restype, fn.argtypes = res, args
        if handle.om_abi_version() != ABI_VERSION:
            raise NativeError("libopenmatch_hip.so ABI version mismatch; rebuild it")
        _lib = handle
    return _lib


def exported_symbols():
    return list(_SIGNATURES)


def check(rc):
    if rc != 0:
        raise NativeError(lib().om_last_error().decode("utf-8", "replace"))


def require_device(*tensors):
    """Every tensor must be a contiguous ROCm device tensor; no silent host path."""
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise NativeError(
                "openmatch_amd runs on MI355X only: got a CPU tensor. Move the model and the batch "
                "to the GPU (model.to('cuda')); there is no CPU / eager fallback.")
        if not t.is_contiguous():
            raise NativeError("non-contiguous tensor passed to the HIP boundary")


def ptr(t):
    return c_void_p(t.data_ptr()) if t is not None else c_void_p(0)


def stream_ptr(device=None):
    return c_void_p(torch.cuda.current_stream(device).cuda_stream)


class Workspace:
    """Grow-only device scratch buffer (256-byte aligned), one per (device, tag)."""
    _pool = {}

    @classmethod
    def get(cls, device, nbytes, tag="default"):
        key = (str(device), tag)
        buf = cls._pool.get(key)
        # the caller gets data_ptr() + off with off up to 255: a buffer is only reused when nbytes fit BEHIND that offset
        if buf is None or buf.numel() - ((-buf.data_ptr()) % 256) < nbytes:
            buf = torch.empty(int(nbytes) + 256, dtype=torch.uint8, device=device)
            cls._pool[key] = buf
        off = (-buf.data_ptr()) % 256
        return buf, buf.data_ptr() + off
