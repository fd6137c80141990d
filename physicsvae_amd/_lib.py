"""ctypes binding of include/pvae.h.  There is no fallback: if the HIP library is missing
or a call fails, a RuntimeError is raised."""
import ctypes as C
import os

# Kernel arguments in device memory: with them in host memory every launch of this latency-bound
# step pays a PCIe read before its first instruction (measured: 128.6 vs 104.7 us per world step).
# ROCm 7 defaults to device kernargs on this GPU; pin it so the behaviour does not depend on that.
# (Read by the HIP runtime when it initialises, i.e. at the first CUDA/HIP call of the process.)
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
if os.environ.get("PVAE_DP_BUCKET_MB", "0") not in ("", "0", "0.0") or int(os.environ.get("WORLD_SIZE", "1") or 1) > 1:
    # overlapped gradient exchange (asked for, or the multi-rank default in the joint phase): keep the
    # library's exchange stream off the NULL stream's hardware queue
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

# PyTorch-ROCm bundles its own HIP runtime (torch/lib/libamdhip64.so).  It must be the first one
# the process loads: our library then binds to the SAME runtime instance (same soname), which is
# what makes torch's device pointers and streams valid inside libpvae.  Loading libpvae first
# would pull in /opt/rocm's runtime instead and leave two runtimes that do not share a context.
import torch  # noqa: F401  (load order matters)

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PVAE_LIB_PATH") or os.path.join(HERE, "libpvae_gfx950.so")   # env: A/B builds

NET_TE, NET_MD, NET_WM, NET_PR, NET_MH = 0, 1, 2, 3, 4
NUM_NETS = 5
NET_NAMES = {NET_TE: "_task_encoder", NET_MD: "_motor_decoder", NET_WM: "_world_model", NET_PR: "_latent_prior",
             NET_MH: "_motor_decoder_helper"}
ARENA_ORDER = (NET_TE, NET_MD, NET_MH, NET_PR, NET_WM)           # pvae_layout.h kArenaOrder
# latent_prior_type (rmt:614-635) -> pvae_config.prior_kind
ACT_KINDS = {"relu": 0, "tanh": 1, "sigmoid": 2, "elu": 3}      # pvae_config.act_kind (get_activation_fn rmt:30-46)
ACT_LINEAR = 4                                                   # per-layer only: no activation after a hidden layer
LAYER_ACTS = dict(ACT_KINDS, linear=ACT_LINEAR)                  # pvae_config.layer_act holds 1 + these
LAYER_ACT_NAMES = {v: k for k, v in LAYER_ACTS.items()}
MAX_HIDDEN = 15
PRIOR_KINDS = {"normal_zero_mean_one_std": 0, "normal_state_mean_one_std": 1, "hypersphere_uniform": 2, False: 3}
PHASE_WORLD, PHASE_JOINT = 0, 1
FLAG_FUSED_ADAM, FLAG_NO_BACKWARD = 1, 2
ABI_VERSION = 12
LOSS_MSE, LOSS_L1 = 0, 1
EXCHANGE_ALLREDUCE, EXCHANGE_SHARDED, EXCHANGE_P2P, EXCHANGE_LOCAL, EXCHANGE_P2P_PUSH = 0, 1, 2, 3, 4
P2P_BLOB_BYTES, P2P_MAX_RANKS = 512, 8


class Config(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "dim_body", "dim_action", "latent", "te_width", "te_depth", "md_width", "md_depth",
        "wm_width", "wm_depth", "max_batch", "lookahead", "prior_kind", "pr_width", "pr_depth", "act_kind")] + [
        ("layer_width", (C.c_int32 * 16) * NUM_NETS), ("layer_act", (C.c_int32 * 16) * NUM_NETS),
        ("te_inputs", C.c_int32), ("md_inputs", C.c_int32),
        ("mh_width", C.c_int32), ("mh_depth", C.c_int32), ("mh_range", C.c_float)]


INPUT_BODY, INPUT_TASK = 1, 2                                     # pvae_config.te_inputs / md_inputs bits (0 = both)
INPUT_BITS = {("body", "task"): 0, ("body",): INPUT_BODY, ("task",): INPUT_TASK}


class LayerInfo(C.Structure):
    _fields_ = [("net", C.c_int32), ("index", C.c_int32), ("n_in", C.c_int32),
                ("n_out", C.c_int32), ("ld", C.c_int32), ("n_out_pad", C.c_int32),
                ("w_offset", C.c_int64), ("b_offset", C.c_int64), ("act", C.c_int32), ("col0", C.c_int32)]


class StepParams(C.Structure):
    _fields_ = [("a_rec_coeff", C.c_float), ("kl_coeff", C.c_float), ("s_rec_coeff", C.c_float),
                ("cycle_coeff", C.c_float), ("lr", C.c_double), ("beta1", C.c_double),
                ("beta2", C.c_double), ("adam_eps", C.c_double), ("adam_t", C.c_int32 * NUM_NETS),
                ("global_rows", C.c_int32), ("rng_seed", C.c_uint64), ("rng_offset", C.c_uint64),
                ("loss_kind", C.c_int32), ("weight_decay", C.c_float)]


_P = C.c_void_p
_SIGS = {
    "pvae_abi_version": (C.c_int, []),
    "pvae_last_error": (C.c_char_p, []),
    "pvae_num_layers": (C.c_int, [C.POINTER(Config)]),
    "pvae_layer": (C.c_int, [C.POINTER(Config), C.c_int, C.POINTER(LayerInfo)]),
    "pvae_arena_floats": (C.c_int64, [C.POINTER(Config)]),
    "pvae_net_segment": (C.c_int, [C.POINTER(Config), C.c_int, C.POINTER(C.c_int64),
                                   C.POINTER(C.c_int64)]),
    "pvae_workspace_bytes": (C.c_size_t, [C.POINTER(Config)]),
    "pvae_workspace_offset": (C.c_int64, [C.POINTER(Config), C.c_int, C.c_int, C.c_int]),
    "pvae_create": (C.c_int, [C.POINTER(Config), C.POINTER(_P)]),
    "pvae_destroy": (None, [_P]),
    "pvae_bind_arenas": (C.c_int, [_P, _P, _P, _P, _P]),
    "pvae_bind_workspace": (C.c_int, [_P, _P, C.c_size_t]),
    "pvae_bind_dataset": (C.c_int, [_P, _P, _P, _P, C.c_int64, C.c_int64]),
    "pvae_bind_dataset_next": (C.c_int, [_P, _P]),
    "pvae_invalidate_staging": (C.c_int, [_P]),
    "pvae_set_direct": (C.c_int, [_P, C.c_int]),
    "pvae_set_option": (C.c_int, [_P, C.c_char_p, C.c_int64]),
    "pvae_direct_active": (C.c_int, [_P, C.c_int, C.c_int32, _P, C.c_int]),
    "pvae_gather": (C.c_int, [_P, C.c_int64, C.c_int32, _P]),
    "pvae_set_batch": (C.c_int, [_P, _P, _P, C.c_int32, _P]),
    "pvae_forward_backward": (C.c_int, [_P, C.c_int, C.c_int32, C.POINTER(StepParams), _P, _P,
                                        C.c_int, _P]),
    "pvae_adam": (C.c_int, [_P, C.c_int, C.POINTER(StepParams), _P]),
    "pvae_forward_seed": (C.c_int, [_P, C.c_int, C.c_int32, C.POINTER(StepParams), _P, _P]),
    "pvae_backward_stage": (C.c_int, [_P, C.c_int, C.c_int32, C.POINTER(StepParams), C.c_int, _P, _P,
                                      C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int),
                                      C.POINTER(C.c_int)]),
    "pvae_adam_segment": (C.c_int, [_P, C.c_int, C.c_int64, C.c_int64, C.POINTER(StepParams), _P]),
    "pvae_backward_plan": (C.c_int, [_P, C.c_int, C.POINTER(StepParams), C.POINTER(C.c_int64), C.POINTER(C.c_int64),
                                     C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_int)]),
    "pvae_comm_unique_id": (C.c_int, [_P]),
    "pvae_comm_init": (C.c_int, [_P, C.c_int, C.c_int, _P]),
    "pvae_comm_destroy": (C.c_int, [_P]),
    "pvae_comm_info": (C.c_int, [_P, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "pvae_comm_config": (C.c_int, [_P, C.c_int64, C.c_int32]),
    "pvae_comm_mode": (C.c_int, [_P, C.c_int]),
    "pvae_p2p_export": (C.c_int, [_P, _P]),
    "pvae_p2p_open": (C.c_int, [_P, C.c_int, C.c_int, _P]),
    "pvae_p2p_close": (C.c_int, [_P]),
    "pvae_p2p_selftest": (C.c_int, [_P, _P]),
    "pvae_p2p_clear_errors": (C.c_int, [_P, _P]),
    "pvae_rollout_is_fused": (C.c_int, []),
    "pvae_rollout_server_start": (C.c_int, [_P, C.c_double, C.c_double, C.c_int]),
    "pvae_rollout_server_infer": (C.c_int, [_P, _P, C.c_int, C.c_uint64, C.c_uint64, C.c_int, _P, _P, _P, C.c_double]),
    "pvae_rollout_server_infer_rows": (C.c_int, [_P, _P, C.c_int32, C.c_int, C.c_uint64, C.c_uint64, C.c_int, _P, _P, _P, C.c_double]),
    "pvae_rollout_server_decode": (C.c_int, [_P, _P, _P, C.c_double]),
    "pvae_rollout_server_stop": (C.c_int, [_P]),
    "pvae_params_changed": (C.c_int, [_P, _P]),
    "pvae_rollout_server_selfbench": (C.c_int, [_P, _P, C.c_int, C.c_int32, _P]),
    "pvae_rollout_server_timeline": (C.c_int, [_P, _P, C.c_int32, C.POINTER(C.c_int32)]),
    "pvae_rollout_server_status": (C.c_int, [_P, C.POINTER(C.c_int32), C.POINTER(C.c_uint32), C.POINTER(C.c_int32)]),
    "pvae_p2p_exchange": (C.c_int, [_P, C.c_int, C.c_int64, C.c_int64, C.POINTER(StepParams), _P]),
    "pvae_p2p_status": (C.c_int, [_P, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_uint32), _P]),
    "pvae_owned_slices": (C.c_int, [_P, C.c_int, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int32),
                                    C.c_int32, C.POINTER(C.c_int32)]),
    "pvae_allreduce_grads": (C.c_int, [_P, C.c_int64, C.c_int64, _P]),
    "pvae_dp_train_step": (C.c_int, [_P, C.c_int, C.c_int64, C.c_int32, C.POINTER(StepParams), _P, _P,
                                     C.c_int64, C.c_int32, _P]),
    "pvae_train_step": (C.c_int, [_P, C.c_int, C.c_int64, C.c_int32, C.POINTER(StepParams), _P, _P,
                                  _P]),
    "pvae_train_step_prefetch": (C.c_int, [_P, C.c_int, C.c_int64, C.c_int32, C.POINTER(StepParams), _P, _P,
                                           C.c_int64, C.c_int32, _P]),
    "pvae_read_tensor": (C.c_int, [_P, C.c_int, _P, C.c_int32, _P]),
    "pvae_infer": (C.c_int, [_P, _P, C.c_int32, _P, C.c_int, C.c_uint64, C.c_uint64, _P, _P, _P, _P]),
    "pvae_infer_logits": (C.c_int, [_P, _P, C.c_int32, _P, C.c_int, C.c_uint64, C.c_uint64, _P, C.c_int32, _P, _P, _P, _P]),
    "pvae_mlp_forward": (C.c_int, [_P, C.c_int32, C.c_int32, C.c_int32, _P, _P, _P, _P, _P, C.c_int32, _P, _P, _P, C.c_int32, _P]),
    "pvae_net_forward": (C.c_int, [_P, C.c_int, _P, C.c_int32, _P, _P]),
    "pvae_reparam": (C.c_int, [_P, _P, C.c_int32, _P, C.c_int, C.c_uint64, C.c_uint64, _P, _P]),
    "pvae_mfma_clock_probe": (C.c_int, [_P, C.c_int64, _P, C.POINTER(C.c_double), C.POINTER(C.c_double), _P]),
    "pvae_profile_enable": (C.c_int, [C.c_int]),
    "pvae_profile_read": (C.c_int, [C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int64),
                                    C.POINTER(C.c_double)]),
    "pvae_gemm_probe": (C.c_int, [C.c_int, _P, C.c_int, _P, C.c_int, _P, C.c_int, _P, C.c_int,
                                  C.c_int, C.c_int, C.c_int, C.c_int, _P]),
}
EXPORTS = tuple(_SIGS)
_lib = None


def load():
    """Load libpvae_gfx950.so (built by physicsvae_amd.build / __graft_entry__.build)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "%s is missing: the HIP hot path is not built (run `python -m physicsvae_amd.build`); "
            "there is no CPU fallback" % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in _SIGS.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if lib.pvae_abi_version() != ABI_VERSION:
        raise RuntimeError("libpvae ABI version mismatch")
    # The library itself reads no environment variable (include/pvae.h pvae_set_option).  The PVAE_* switches of the tests
    # and the A/B scripts under tools/ are mapped onto that call here, once per process: kernel-geometry switches.
    for env, (key, conv) in PROCESS_OPTIONS.items():
        if env in os.environ:
            rc = lib.pvae_set_option(None, key.encode(), conv(os.environ[env]))
            if rc < 0:
                raise RuntimeError("pvae_set_option(%s) failed: %s" % (key, lib.pvae_last_error().decode()))
    _lib = lib
    return lib


def _flag(v):
    return 0 if v[:1] == "0" else 1


# environment variable -> (option name, value conversion); see include/pvae.h pvae_set_option
PROCESS_OPTIONS = {
    "PVAE_KROT": ("krot", _flag), "PVAE_ROWXCD": ("rowxcd", _flag), "PVAE_WS64": ("ws64", _flag),
    "PVAE_WS6464": ("ws6464", _flag), "PVAE_WS6464_ROWS": ("ws6464_rows", _flag), "PVAE_PAIR64": ("pair64", _flag),
    "PVAE_DGRAD16": ("dgrad16", _flag), "PVAE_WGRAD32": ("wgrad32", int), "PVAE_LOOK_PAIR": ("look_pair", _flag),
    "PVAE_ROLLOUT_FUSED": ("rollout_fused", _flag),
}
CONTEXT_OPTIONS = {
    "PVAE_PAIR": ("pair", _flag), "PVAE_DEFER_ADAM": ("defer_adam", _flag), "PVAE_SAME_LAYER": ("same_layer", _flag),
    "PVAE_FOLD_SAMPLER": ("fold_sampler", _flag), "PVAE_DIRECT": ("direct", _flag),
    "PVAE_P2P_TIMEOUT_MS": ("p2p_timeout_ms", int), "PVAE_P2P_SELFTEST_FLAGS_ONLY": ("p2p_selftest_flags_only", _flag),
    "PVAE_SERVER_MAILBOX": ("server_mailbox", lambda v: {"h": 1, "d": 2}.get(v[:1], 0)),
}


def apply_context_options(lib, ctx):
    """The per-context switches of the environment, applied to a freshly created context."""
    for env, (key, conv) in CONTEXT_OPTIONS.items():
        if env in os.environ:
            check(lib.pvae_set_option(ctx, key.encode(), conv(os.environ[env])), "pvae_set_option(%s)" % key)


def check(rc, what=""):
    if rc < 0:
        msg = load().pvae_last_error().decode("utf-8", "replace")
        raise RuntimeError("libpvae %s failed (%d): %s" % (what, rc, msg))
    return rc
