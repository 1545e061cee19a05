"""ctypes binding of include/bsarec_hip.h.  Fails loudly: there is no CPU or PyTorch fallback."""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("BSAREC_LIB") or os.path.join(HERE, "libbsarec_hip.so")   # BSAREC_LIB: another build of the same ABI
MAX_LAYERS = 16
ABI_VERSION = 8

(BUF_LAYER_OUT, BUF_LOGITS, BUF_LOSS, BUF_DSP, BUF_HMIX, BUF_PROBS, BUF_DLAYER_IN, BUF_LOSS_ROWS, BUF_CTX,
 BUF_DLOGITS) = range(10)
K_NONE, K_FFN1, K_FFN2, K_QKV, K_LOGITS, K_DU, K_DW1, K_FUSED_FWD, K_FUSED_BWD = range(9)

LAYER_FIELDS = ["sqrt_beta", "filter_ln_w", "filter_ln_b", "query_w", "query_b", "key_w", "key_b", "value_w", "value_b",
                "dense_w", "dense_b", "attn_ln_w", "attn_ln_b", "ffn1_w", "ffn1_b", "ffn2_w", "ffn2_b", "ffn_ln_w",
                "ffn_ln_b"]
# C field name -> state_dict suffix under item_encoder.blocks.{l}.
LAYER_KEYS = {
    "sqrt_beta": "layer.filter_layer.sqrt_beta",
    "filter_ln_w": "layer.filter_layer.LayerNorm.weight", "filter_ln_b": "layer.filter_layer.LayerNorm.bias",
    "query_w": "layer.attention_layer.query.weight", "query_b": "layer.attention_layer.query.bias",
    "key_w": "layer.attention_layer.key.weight", "key_b": "layer.attention_layer.key.bias",
    "value_w": "layer.attention_layer.value.weight", "value_b": "layer.attention_layer.value.bias",
    "dense_w": "layer.attention_layer.dense.weight", "dense_b": "layer.attention_layer.dense.bias",
    "attn_ln_w": "layer.attention_layer.LayerNorm.weight", "attn_ln_b": "layer.attention_layer.LayerNorm.bias",
    "ffn1_w": "feed_forward.dense_1.weight", "ffn1_b": "feed_forward.dense_1.bias",
    "ffn2_w": "feed_forward.dense_2.weight", "ffn2_b": "feed_forward.dense_2.bias",
    "ffn_ln_w": "feed_forward.LayerNorm.weight", "ffn_ln_b": "feed_forward.LayerNorm.bias",
}
OPTIONAL_LAYER_KEYS = {"filter_cw": "layer.filter_layer.complex_weight"}      # sibling model FMLPRec only
TOP_KEYS = {"item_emb": "item_embeddings.weight", "pos_emb": "position_embeddings.weight",
            "ln_w": "LayerNorm.weight", "ln_b": "LayerNorm.bias"}


class Config(C.Structure):
    _fields_ = [("batch", C.c_int), ("seq_len", C.c_int), ("hidden", C.c_int), ("heads", C.c_int), ("layers", C.c_int),
                ("item_size", C.c_int), ("cutoff_bins", C.c_int), ("alpha", C.c_float), ("ln_eps", C.c_float),
                ("p_hidden", C.c_float), ("p_attn", C.c_float), ("filter_kind", C.c_int),
                # per-plan options, 0 = default (include/bsarec_hip.h)
                ("hidden_act", C.c_int), ("storage", C.c_int), ("no_fused", C.c_int), ("no_prune_top", C.c_int),
                ("dw_tiled", C.c_int), ("splits", C.c_int), ("top_slabs", C.c_int), ("separate_embed", C.c_int),
                ("separate_top", C.c_int), ("chain_kernels", C.c_int), ("x3_products", C.c_int)]


OPTION_FIELDS = ("storage", "no_fused", "no_prune_top", "dw_tiled", "splits", "top_slabs", "separate_embed", "separate_top", "chain_kernels", "x3_products")
HIDDEN_ACTS = {"gelu": 0, "relu": 1, "swish": 2, "tanh": 3, "sigmoid": 4}      # src/model/_modules.py:38-45

# Plan options the HOST gives to plans it creates from now on.  The C ABI has no process-wide state: these are Python
# defaults (test / bench shims and the environment knobs of INTEGRATION.md), copied into bsarec_config_t per plan.
_defaults = {k: 0 for k in OPTION_FIELDS}


def _env_defaults():
    e = os.environ
    d = {}
    if e.get("BSAREC_DW") == "tiled":
        d["dw_tiled"] = 1
    if e.get("BSAREC_PRUNE_TOP") == "0":
        d["no_prune_top"] = 1
    if e.get("BSAREC_EMBED_IN_BLOCK") == "0":
        d["separate_embed"] = 1
    if e.get("BSAREC_TOP_TAIL") == "0":
        d["separate_top"] = 1
    if e.get("BSAREC_FUSED") == "0":
        d["no_fused"] = 1
    if e.get("BSAREC_BLOCK_KERNELS") == "chain":
        d["chain_kernels"] = 1
    if e.get("BSAREC_PRODUCTS") == "bf16x3":
        d["x3_products"] = 1
    for env, key, hi in (("BSAREC_TOP_SLABS", "top_slabs", 16), ("BSAREC_SPLITS", "splits", 1024)):
        if e.get(env, "").isdigit() and 1 <= int(e[env]) <= hi:
            d[key] = int(e[env])
    if e.get("BSAREC_STORAGE") == "bf16":
        d["storage"] = 1
    return d


def set_default_options(**kw):
    """Test / bench shim: options of the plans created after this call (e.g. ``no_prune_top=1``).  Returns the
    previous values of the keys it changed."""
    old = {}
    for k, v in kw.items():
        if k not in _defaults:
            raise KeyError(k)
        old[k] = _defaults[k]
        _defaults[k] = int(v)
    return old


def default_options():
    d = dict(_defaults)
    d.update(_env_defaults())
    return d


class Layer(C.Structure):
    _fields_ = [(f, C.c_void_p) for f in LAYER_FIELDS] + [("filter_cw", C.c_void_p)]


class Tensors(C.Structure):
    _fields_ = [("item_emb", C.c_void_p), ("pos_emb", C.c_void_p), ("ln_w", C.c_void_p), ("ln_b", C.c_void_p),
                ("layer", Layer * MAX_LAYERS)]


class Adam(C.Structure):
    """bsarec_adam_t"""
    _fields_ = [("params", C.c_void_p), ("grads", C.c_void_p), ("exp_avg", C.c_void_p), ("exp_avg_sq", C.c_void_p),
                ("n", C.c_long), ("lr", C.c_float), ("beta1", C.c_float), ("beta2", C.c_float), ("eps", C.c_float),
                ("weight_decay", C.c_float), ("grad_scale", C.c_float), ("shadow_bf16", C.c_void_p), ("shadow_from", C.c_long),
                ("grads2", C.c_void_p), ("grads2_n", C.c_long), ("n_grad_srcs", C.c_int), ("grad_srcs", C.c_void_p * 8)]


class Comm(C.Structure):
    """bsarec_comm_t (include/bsarec_comm.h)"""
    _fields_ = [("rank", C.c_int), ("world", C.c_int), ("flags", C.c_void_p * 8), ("epoch", C.c_void_p),
                ("error", C.c_void_p), ("timeout_ms", C.c_int)]


HOOK = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p)          # bsarec_hook_t


EXPORTS = {
    "bsarec_abi_version": (C.c_int, []),
    "bsarec_workspace_bytes": (C.c_size_t, [C.POINTER(Config)]),
    "bsarec_plan_create": (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(Config), C.POINTER(Tensors), C.POINTER(Tensors),
                                     C.POINTER(Tensors), C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p]),
    "bsarec_shadow_refresh": (C.c_int, [C.c_void_p, C.c_void_p]),
    "bsarec_plan_set_dense_grad_hook": (C.c_int, [C.c_void_p, HOOK, C.c_void_p, C.c_void_p]),
    "bsarec_buffer_is_bf16": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "bsarec_plan_is_fused": (C.c_int, [C.c_void_p]),
    "bsarec_config_is_fused": (C.c_int, [C.c_void_p]),
    "bsarec_plan_destroy": (None, [C.c_void_p]),
    "bsarec_buffer_offset": (C.c_long, [C.c_void_p, C.c_int, C.c_int]),
    "bsarec_step_begin": (C.c_int, [C.c_void_p, C.c_void_p]),
    "bsarec_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "bsarec_forward_last": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "bsarec_loss": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "bsarec_loss_bce": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "bsarec_loss_logsig": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "bsarec_logits": (C.c_int, [C.c_void_p, C.c_void_p]),
    "bsarec_backward": (C.c_int, [C.c_void_p, C.c_void_p]),
    "bsarec_backward_seq": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "bsarec_backward_seq_multi": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "bsarec_adam_step": (C.c_int, [C.POINTER(Adam), C.c_void_p, C.c_void_p]),
    "bsarec_train_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(Adam), C.c_void_p]),
    "bsarec_gather_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_long, C.c_void_p, C.c_int, C.c_int,
                                      C.c_void_p, C.c_void_p, C.c_void_p]),
    "bsarec_train_step_indexed": (C.c_int, [C.c_void_p] * 4 + [C.c_long] + [C.c_void_p] * 3 + [C.POINTER(Adam), C.c_void_p]),
    "bsarec_grad_step_indexed": (C.c_int, [C.c_void_p] * 4 + [C.c_long] + [C.c_void_p] * 3 + [C.c_float] * 3 + [C.c_void_p]),
    "bsarec_adam_apply": (C.c_int, [C.POINTER(Adam), C.c_void_p, C.c_void_p]),
    "bsarec_mask_seen": (C.c_int, [C.c_void_p, C.c_long, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "bsarec_topk_seen": (C.c_int, [C.c_void_p, C.c_long, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                   C.c_void_p, C.c_void_p]),
    "bsarec_freq_layer_fwd": (C.c_int, [C.c_void_p] * 5 + [C.c_int] * 4 + [C.c_float, C.c_float, C.c_void_p, C.c_int,
                                                                          C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "bsarec_freq_layer_bwd_scratch_floats": (C.c_long, [C.c_int, C.c_int, C.c_int]),
    "bsarec_freq_layer_bwd": (C.c_int, [C.c_void_p] * 7 + [C.c_int] * 4 + [C.c_float, C.c_void_p, C.c_int] +
                              [C.c_void_p] * 6),
    "bsarec_profile_select": (C.c_int, [C.c_void_p, C.c_int]),
    "bsarec_debug_stamps": (C.c_int, [C.c_void_p, C.c_void_p]),
    "bsarec_profile_read": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int)]),
    "bsarec_profile_event_overhead": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_double)]),
}

# include/bsarec_comm.h
COMM_EXPORTS = {
    "bsarec_comm_alloc": (C.c_int, [C.POINTER(C.c_void_p), C.c_size_t, C.c_int]),
    "bsarec_comm_free": (C.c_int, [C.c_void_p]),
    "bsarec_comm_export": (C.c_int, [C.c_void_p, C.c_char_p]),
    "bsarec_comm_import": (C.c_int, [C.c_char_p, C.POINTER(C.c_void_p)]),
    "bsarec_comm_release": (C.c_int, [C.c_void_p]),
    "bsarec_comm_barrier": (C.c_int, [C.POINTER(Comm), C.c_void_p]),
}

# include/bsarec_shard.h
PTRS8 = C.c_void_p * 8
SHARD_EXPORTS = {
    "bsarec_shard_gather_rows": (C.c_int, [C.c_void_p, C.c_long, C.POINTER(PTRS8), C.c_int, C.c_long, C.c_long, C.c_int,
                                           C.c_void_p, C.c_void_p, C.c_void_p]),
    "bsarec_shard_logits": (C.c_int, [C.c_void_p, C.c_long, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_long,
                                      C.c_void_p]),
    "bsarec_shard_ce_stats": (C.c_int, [C.c_void_p, C.c_long, C.c_int, C.c_int, C.c_void_p, C.c_long, C.c_long, C.c_void_p,
                                        C.c_void_p]),
    "bsarec_shard_ce_grad": (C.c_int, [C.c_void_p, C.c_long, C.c_int, C.c_int, C.c_void_p, C.c_long, C.c_long, C.c_void_p,
                                       C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "bsarec_shard_head_bwd_scratch_floats": (C.c_long, [C.c_int, C.c_int, C.c_int]),
    "bsarec_shard_head_bwd": (C.c_int, [C.c_void_p, C.c_long, C.c_int, C.c_int, C.c_void_p, C.c_long, C.c_void_p, C.c_int,
                                        C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "bsarec_shard_scatter_rows": (C.c_int, [C.c_void_p, C.c_long, C.c_int, C.POINTER(PTRS8), C.c_long, C.c_long, C.c_long,
                                            C.c_int, C.c_void_p, C.c_void_p]),
}

_lib = None


def source_sha16() -> str:
    """First 16 hex digits of the sha256 over the library's sources (csrc/ + the header): the tag that ties an offline
    rocprofv3 profile under profiles/ to the build it was taken with."""
    import hashlib
    h = hashlib.sha256()
    src = os.path.join(HERE, "csrc")
    for f in sorted(os.listdir(src)) + [os.path.join(os.path.dirname(HERE), "include", "bsarec_hip.h"),
                                         os.path.join(HERE, "build.py")]:          # (build.py: the compiler flags)
        with open(f if os.path.isabs(f) else os.path.join(src, f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def load():
    """dlopen the HIP library and bind every symbol of the header.  Raises if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build it with `python -m bsarec_amd.build` (hipcc, gfx950). "
            "bsarec_amd has no fallback path.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in list(EXPORTS.items()) + list(COMM_EXPORTS.items()) + list(SHARD_EXPORTS.items()):
        fn = getattr(lib, name)       # AttributeError if the library does not export it
        fn.restype = res
        fn.argtypes = args
    if lib.bsarec_abi_version() != ABI_VERSION:
        raise RuntimeError("libbsarec_hip.so ABI version mismatch: rebuild with `python -m bsarec_amd.build --force`")
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        kind = "invalid argument / unsupported shape" if rc < 0 else "hipError_t"
        raise RuntimeError(f"{what} failed: {kind} {rc}")
