"""ctypes binding of libultr_hip.so (the C ABI declared in include/ultr_hip.h).

The reference-side stub shown in INTEGRATION.md is this file minus the conveniences.
Fails loudly when the library is missing — there is no fallback path.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libultr_hip.so")

ULTR_MAX_HIDDEN = 7
COMM_HANDLE_BYTES, COMM_MAX_WORLD = 64, 8
ACT = {"elu": 0, "relu": 1, "tanh": 2, "sigmoid": 3}  # base_ranking_model.py:63-69 ("selu" raises in the reference)
ATTN_DTYPE = {"fp32": 0, "fp16": 1}
ALGO_SOFTMAX, ALGO_DLA, ALGO_PAIRDEBIAS, ALGO_LAMBDARANK, ALGO_REGEM = 0, 1, 2, 3, 4
OPT_ADAGRAD, OPT_SGD = 0, 1

c_i32, c_i64, c_f32, c_vp = ctypes.c_int32, ctypes.c_int64, ctypes.c_float, ctypes.c_void_p


class DnnDesc(ctypes.Structure):
    _fields_ = [("feature_size", c_i32), ("n_hidden", c_i32), ("hidden", c_i32 * ULTR_MAX_HIDDEN), ("activation", c_i32),
                ("flags", c_i32)]


MODEL_FP32_PRODUCTS = 1  # ultr_dnn_desc / ultr_setrank_desc ::flags (include/ultr_hip.h, ABI 6)
ABI_VERSION = 8          # include/ultr_hip.h: ULTR_ABI_VERSION - load() refuses a library that reports another one


class UpdateDesc(ctypes.Structure):
    _fields_ = [("algo", c_i32), ("optimizer", c_i32), ("list_size", c_i32), ("logits_to_prob", c_i32),
                ("n_params", c_i64), ("learning_rate", c_f32), ("max_gradient_norm", c_f32), ("adagrad_eps", c_f32),
                ("ranker_loss_weight", c_f32), ("propensity_learning_rate", c_f32), ("em_step_size", c_f32),
                ("regulation_p", c_f32), ("l2_loss", c_f32), ("guard", c_vp), ("host_scalars", c_vp),
                ("seq", ctypes.c_uint32), ("pad_", ctypes.c_uint32), ("range_flag", c_vp)]


class ClickArgs(ctypes.Structure):  # ultr_click_args (ABI 7)
    _fields_ = [("lists", c_vp), ("labels", c_vp), ("n_queries", c_i64), ("n_docs", c_i64), ("exam_prob", c_vp), ("click_prob", c_vp),
                ("lmax", c_i32), ("n_exam", c_i32), ("n_rel", c_i32), ("click_model", c_i32), ("seed", ctypes.c_uint64),
                ("step", ctypes.c_uint64), ("batch", c_i32), ("list_size", c_i32), ("max_tries", c_i32), ("pad_", c_i32),
                ("docids", c_vp), ("clicks", c_vp), ("query_idx", c_vp)]


class SetRankDesc(ctypes.Structure):
    _fields_ = [("feature_size", c_i32), ("d_model", c_i32), ("num_heads", c_i32), ("num_layers", c_i32), ("dff", c_i32),
                ("attention_dtype", c_i32), ("flags", c_i32)]


class StepArgs(ctypes.Structure):
    _fields_ = [("desc", ctypes.POINTER(DnnDesc)), ("upd", ctypes.POINTER(UpdateDesc)), ("params", c_vp), ("wt", c_vp),
                ("state", c_vp), ("aux", c_vp), ("features", c_vp), ("docids", c_vp), ("labels", c_vp), ("pw", c_vp),
                ("ipw_table", c_vp), ("scores", c_vp), ("dscores", c_vp), ("saved", c_vp), ("loss_ws", c_vp),
                ("bwd_ws", c_vp), ("grads", c_vp), ("scalars", c_vp), ("n_docs", c_i64), ("n_ipw", c_i32),
                ("batch", c_i32), ("list_size", c_i32), ("batch_total", c_i32), ("skip_update", c_i32), ("sigma", c_f32),
                ("uniforms", c_vp), ("rng_seed", ctypes.c_uint64), ("rng_step", ctypes.c_uint64),
                ("comm", c_vp), ("comm_step", ctypes.c_uint64)]


# name -> (restype, argtypes); must list EVERY symbol include/ultr_hip.h declares
SIGNATURES = {
    "ultr_abi_version": (c_i32, []),
    "ultr_dnn_param_count": (c_i64, [ctypes.POINTER(DnnDesc)]),
    "ultr_dnn_param_offsets": (c_i32, [ctypes.POINTER(DnnDesc), ctypes.POINTER(c_i64)]),
    "ultr_dnn_saved_bytes": (c_i64, [ctypes.POINTER(DnnDesc), c_i64]),
    "ultr_dnn_bwd_workspace_bytes": (c_i64, [ctypes.POINTER(DnnDesc), c_i64]),
    "ultr_step_tail_floats": (c_i64, [c_i32]),
    "ultr_loss_workspace_bytes": (c_i64, [c_i64, c_i32]),
    "ultr_loss_part_count": (c_i64, [c_i64]),
    "ultr_dnn_forward": (c_i32, [ctypes.POINTER(DnnDesc), c_vp, c_vp, c_vp, c_i64, c_vp, c_i32, c_i32, c_vp, c_vp, c_vp]),
    "ultr_dnn_wt_floats": (c_i64, [ctypes.POINTER(DnnDesc)]),
    "ultr_dnn_build_wt": (c_i32, [ctypes.POINTER(DnnDesc), c_vp, c_vp, c_vp]),
    "ultr_dnn_wt_range": (c_i32, [ctypes.POINTER(DnnDesc), c_vp, c_vp]),
    "ultr_dnn_forward_tile_rows": (c_i32, [ctypes.POINTER(DnnDesc), c_i64, c_i32]),
    "ultr_dnn_backward_tile_rows": (c_i32, [ctypes.POINTER(DnnDesc), c_i64]),
    "ultr_dnn_backward": (c_i32, [ctypes.POINTER(DnnDesc), c_vp, c_vp, c_i64, c_vp, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp,
                                  c_vp, c_vp]),
    "ultr_dnn_backward_softmax": (c_i32, [ctypes.POINTER(DnnDesc), c_vp, c_vp, c_i64, c_vp, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp,
                                          c_vp, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "ultr_grad_sumsq": (c_i32, [c_vp, c_i64, c_i32, c_vp, c_vp]),
    "ultr_softmax_ce": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_vp, c_vp, c_vp]),
    "ultr_dla_loss": (c_i32, [c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_vp, c_vp, c_vp]),
    "ultr_pairdebias_loss": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_vp, c_vp, c_vp]),
    "ultr_lambdarank_loss": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_f32, c_i32, c_i32, c_vp, c_vp, c_vp]),
    "ultr_regem_loss": (c_i32, [c_vp, c_vp, c_vp, c_vp, ctypes.c_uint64, ctypes.c_uint64, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp]),
    "ultr_setrank_param_count": (c_i64, [ctypes.POINTER(SetRankDesc)]),
    "ultr_setrank_saved_bytes": (c_i64, [ctypes.POINTER(SetRankDesc), c_i64]),
    "ultr_setrank_workspace_bytes": (c_i64, [ctypes.POINTER(SetRankDesc), c_i64]),
    "ultr_setrank_range_flag_offset": (c_i64, [ctypes.POINTER(SetRankDesc), c_i64]),
    "ultr_setrank_forward": (c_i32, [ctypes.POINTER(SetRankDesc), c_vp, c_vp, c_i64, c_vp, c_i32, c_i32, c_vp, c_vp, c_vp]),
    "ultr_setrank_backward": (c_i32, [ctypes.POINTER(SetRankDesc), c_vp, c_i32, c_i32, c_vp, c_vp, c_vp, c_i32, c_vp, c_vp,
                                      c_vp]),
    "ultr_apply_update": (c_i32, [ctypes.POINTER(UpdateDesc), ctypes.POINTER(DnnDesc), c_vp, c_vp, c_vp, c_vp, c_vp, c_vp,
                                  c_vp, c_vp]),
    "ultr_train_step": (c_i32, [ctypes.POINTER(StepArgs), c_vp]),
    "ultr_click_batch": (c_i32, [c_vp, c_vp, c_i64, c_i32, c_i64, c_vp, c_i32, c_vp, c_i32, c_i32, ctypes.c_uint64, ctypes.c_uint64,
                                 c_i32, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp]),
    "ultr_click_batch_args": (c_i32, [c_vp, c_vp]),
    "ultr_feed_train_step": (c_i32, [c_vp, c_vp, c_vp]),
    "ultr_comm_create": (c_i32, [c_i32, c_i32, c_i64, ctypes.POINTER(c_vp)]),
    "ultr_comm_export": (c_i32, [c_vp, c_vp]),
    "ultr_comm_import": (c_i32, [c_vp, c_i32, c_vp]),
    "ultr_comm_allreduce": (c_i32, [c_vp, ctypes.c_uint64, c_vp, c_i64, c_i64, c_vp, c_vp, c_i32, c_vp]),
    "ultr_comm_status": (c_i32, [c_vp, c_vp]),
    "ultr_comm_destroy": (c_i32, [c_vp]),
    "ultr_config_reload": (c_i32, []),
    "ultr_prof_enable": (c_i32, [ctypes.c_uint32, c_i32]),
    "ultr_prof_set_stride": (c_i32, [c_i32]),
    "ultr_prof_collect": (c_i32, [ctypes.POINTER(ctypes.c_double), ctypes.POINTER(c_i64)]),
    "ultr_ndcg": (c_i32, [c_vp, c_vp, c_vp, c_i64, c_i32, c_i32, ctypes.POINTER(c_i32), c_i32, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "ultr_ndcg_report": (c_i32, [c_vp, c_vp, c_vp, c_i64, c_i32, c_i32, ctypes.POINTER(c_i32), c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp,
                                 ctypes.c_uint32, c_vp]),
    "ultr_dnn_forward_ndcg": (c_i32, [ctypes.POINTER(DnnDesc), c_vp, c_vp, c_vp, c_i64, c_vp, c_vp, c_i32, c_i32, c_vp, ctypes.POINTER(c_i32),
                                      c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, ctypes.c_uint32, c_vp]),
}

_LIB = None


def load(path=None):
    """dlopen libultr_hip.so and type every entry point.  Raises if it is missing."""
    global _LIB
    if _LIB is not None and path is None:
        return _LIB
    p = path or os.environ.get("ULTR_HIP_LIB") or LIB_PATH  # ULTR_HIP_LIB: experiment builds (tools/ab_build.sh)
    if not os.path.exists(p):
        raise RuntimeError(
            "libultr_hip.so not found at %s — build it first: python -c 'import __graft_entry__ as g; g.build()' "
            "(ultra_pytorch_amd has no CPU fallback)" % p)
    try:  # make sure the HIP runtime the process will use is torch's (same SONAME libamdhip64.so.7)
        import torch  # noqa: F401
    except Exception:  # pragma: no cover - symbol checks work without torch
        pass
    lib = ctypes.CDLL(p)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the .so does not export it
        fn.restype = res
        fn.argtypes = args
    got = int(lib.ultr_abi_version())
    if got != ABI_VERSION:
        # a stale or experimental build (ULTR_HIP_LIB) that happens to export every symbol would be called with shifted arguments
        raise RuntimeError("%s reports ABI %d, this binding is written for ABI %d: rebuild the library "
                           "(python -c 'import __graft_entry__ as g; g.build()')" % (p, got, ABI_VERSION))
    if path is None:
        _LIB = lib
    return lib


def make_desc(feature_size, hidden, activation="elu"):
    hidden = list(hidden or [])
    if len(hidden) > ULTR_MAX_HIDDEN:
        raise ValueError("at most %d hidden layers are supported" % ULTR_MAX_HIDDEN)
    if activation not in ACT:
        raise ValueError("activation_func must be one of %s (got %r)" % (sorted(ACT), activation))
    d = DnnDesc()
    d.feature_size = int(feature_size)
    d.n_hidden = len(hidden)
    for i, h in enumerate(hidden):
        d.hidden[i] = int(h)
    d.activation = ACT[activation]
    return d


class UltrHipError(RuntimeError):
    pass


def check(rc, what):
    if rc != 0:
        kind = {-1: "bad argument", -2: "unsupported shape", -3: "workspace too small", -4: "peer wait timed out"}.get(rc, "hipError_t %d" % rc)
        raise UltrHipError("%s failed: %s" % (what, kind))
