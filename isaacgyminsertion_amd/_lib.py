"""ctypes binding of libigi_hip.so (C ABI declared in include/igi_ppo.h).

The library is the product: if it is missing or fails to load we raise -- there is no
CPU / eager fallback anywhere in this package.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libigi_hip.so")

IGI_MAX_LAYERS = 4
IGI_MAX_ACT = 8
IGI_STATS_PER_STEP = 8
ABI_VERSION = 3
IGI_E_BADARG, IGI_E_WORKSPACE, IGI_E_UNSUPPORTED, IGI_E_CALLBACK, IGI_E_COMM = -1, -2, -3, -5, -6   # include/igi_ppo.h

EPI_STORE, EPI_BIAS_TANH, EPI_TANHGRAD, EPI_BIAS = 0, 1, 2, 3


class TeacherCfg(C.Structure):
    """struct igi_teacher_cfg"""
    _fields_ = [
        ("obs_dim", C.c_int32), ("priv_dim", C.c_int32), ("act_dim", C.c_int32),
        ("n_priv_layers", C.c_int32), ("priv_units", C.c_int32 * IGI_MAX_LAYERS),
        ("n_layers", C.c_int32), ("units", C.c_int32 * IGI_MAX_LAYERS),
        ("num_envs", C.c_int32), ("horizon", C.c_int32), ("mini_epochs", C.c_int32),
        ("_pad0", C.c_int32),
        ("gamma", C.c_double), ("tau", C.c_double),
        ("lr", C.c_double), ("beta1", C.c_double), ("beta2", C.c_double), ("adam_eps", C.c_double),
        ("e_clip", C.c_float), ("critic_coef", C.c_float), ("entropy_coef", C.c_float),
        ("bounds_loss_coef", C.c_float), ("grad_norm", C.c_float), ("rms_eps", C.c_float),
    ]


class Rollout(C.Structure):
    """struct igi_rollout"""
    _fields_ = [(k, C.c_void_p) for k in
                ("obses", "priv_info", "rewards", "values", "neglogpacs", "dones", "actions", "mus",
                 "sigmas", "last_values")]


class TeacherState(C.Structure):
    """struct igi_teacher_state"""
    _fields_ = [(k, C.c_void_p) for k in
                ("params", "grads", "adam_m", "adam_v", "rms_obs", "rms_priv", "rms_value", "perm",
                 "returns_raw", "advantages", "values_n", "returns_n", "mus_w", "sigmas_w", "stats",
                 "workspace")] + [("workspace_bytes", C.c_size_t)]


class TactileCfg(C.Structure):
    """struct igi_tactile_cfg"""
    _fields_ = [("batch", C.c_int32), ("height", C.c_int32), ("width", C.c_int32), ("latent_dim", C.c_int32)]


class TokenCfg(C.Structure):
    """struct igi_token_cfg"""
    _fields_ = [("batch", C.c_int32), ("seq", C.c_int32), ("d_model", C.c_int32), ("nhead", C.c_int32),
                ("ff", C.c_int32), ("layers", C.c_int32), ("dropout", C.c_float), ("training", C.c_int32)]


class DepthCfg(C.Structure):
    """struct igi_depth_cfg"""
    _fields_ = [("batch", C.c_int32), ("latent_dim", C.c_int32)]


class ProfEntry(C.Structure):
    """struct igi_prof_entry"""
    _fields_ = [("name", C.c_char_p), ("launches", C.c_int64), ("total_ms", C.c_double),
                ("flops", C.c_double), ("bytes", C.c_double)]


_EXPORTS = {
    # name: (restype, argtypes)
    "igi_abi_version": (C.c_int, []),
    "igi_last_error": (C.c_char_p, []),
    "igi_build_info": (C.c_char_p, []),
    "igi_gemm_set_bf16_inputs": (C.c_int, [C.c_int]),
    "igi_gemm_set_bf16x3": (C.c_int, [C.c_int]),
    "igi_teacher_set_norm_fusion": (C.c_int, [C.c_int]),
    "igi_teacher_set_latz_fuse": (C.c_int, [C.c_int]),
    "igi_prof_enable": (C.c_int, [C.c_int]),
    "igi_prof_read": (C.c_int, [C.POINTER(ProfEntry), C.c_int]),
    "igi_mfma_peak_probe": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "igi_gemm_f32": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int,
                               C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                               C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "igi_level_backward_parts": (C.c_int, [C.c_int64, C.c_int, C.c_int]),
    "igi_level_backward": (C.c_int, [C.c_void_p] * 6 + [C.c_int64, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "igi_level_backward_below": (C.c_int, [C.c_void_p] * 8 + [C.c_int64, C.c_int, C.c_int, C.c_void_p]),
    "igi_rms_workspace_bytes": (C.c_size_t, [C.c_int64, C.c_int]),
    "igi_rms_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_float,
                                  C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]),
    "igi_rollout_act_store": (C.c_int, [C.c_int64, C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 7 + [C.c_float] +
                              [C.c_void_p] * 10),
    "igi_rollout_env_store": (C.c_int, [C.c_int64] + [C.c_void_p] * 5 + [C.c_float, C.c_int] + [C.c_void_p] * 7),
    "igi_bc_loss_workspace_bytes": (C.c_size_t, []),
    "igi_bc_loss": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p,
                              C.c_void_p, C.c_size_t, C.c_void_p]),
    "igi_teacher_param_count": (C.c_int64, [C.POINTER(TeacherCfg)]),
    "igi_teacher_param_offsets": (C.c_int, [C.POINTER(TeacherCfg), C.POINTER(C.c_int64),
                                            C.POINTER(C.c_int64), C.c_int]),
    "igi_teacher_workspace_bytes": (C.c_size_t, [C.POINTER(TeacherCfg)]),
    "igi_teacher_prepare": (C.c_int, [C.POINTER(TeacherCfg), C.POINTER(Rollout),
                                      C.POINTER(TeacherState), C.c_int, C.c_void_p]),
    "igi_teacher_fwd_bwd": (C.c_int, [C.POINTER(TeacherCfg), C.POINTER(Rollout),
                                      C.POINTER(TeacherState), C.c_int, C.c_int, C.c_void_p]),
    "igi_teacher_fwd_bwd_phase": (C.c_int, [C.POINTER(TeacherCfg), C.POINTER(Rollout),
                                            C.POINTER(TeacherState), C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "igi_teacher_grad_buckets": (C.c_int, [C.POINTER(TeacherCfg), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "igi_teacher_apply": (C.c_int, [C.POINTER(TeacherCfg), C.POINTER(TeacherState), C.c_int,
                                    C.c_int64, C.c_float, C.c_void_p]),
    "igi_teacher_update": (C.c_int, [C.POINTER(TeacherCfg), C.POINTER(Rollout),
                                     C.POINTER(TeacherState), C.c_int64, C.c_void_p]),
    "igi_teacher_update_dp": (C.c_int, [C.POINTER(TeacherCfg), C.POINTER(Rollout), C.POINTER(TeacherState),
                                        C.c_int64, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]),
    "igi_comm_unique_id": (C.c_int, [C.c_void_p]),
    "igi_comm_create": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "igi_comm_destroy": (C.c_int, [C.c_void_p]),
    "igi_comm_rank": (C.c_int, [C.c_void_p]),
    "igi_comm_world": (C.c_int, [C.c_void_p]),
    "igi_comm_count": (C.c_int, [C.c_void_p]),
    "igi_rccl_version": (C.c_int, []),
    "igi_comm_all_reduce_async_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "igi_comm_join": (C.c_int, [C.c_void_p, C.c_void_p]),
    "igi_comm_last_error": (C.c_char_p, [C.c_void_p]),
    "igi_comm_all_reduce_sum_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "igi_comm_broadcast": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p]),
    "igi_teacher_update_dp_rccl": (C.c_int, [C.POINTER(TeacherCfg), C.POINTER(Rollout), C.POINTER(TeacherState),
                                             C.c_int64, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "igi_teacher_infer": (C.c_int, [C.POINTER(TeacherCfg), C.POINTER(TeacherState), C.c_void_p,
                                    C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                    C.c_void_p]),
    "igi_rollout_policy_step": (C.c_int, [C.POINTER(TeacherCfg), C.POINTER(TeacherState), C.c_void_p, C.c_void_p,
                                          C.c_int64, C.c_int] + [C.c_void_p] * 11 + [C.c_void_p]),
    "igi_clip_adam_workspace_bytes": (C.c_size_t, []),
    "igi_clip_adam": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_float, C.c_double,
                                C.c_double, C.c_double, C.c_double, C.c_int64, C.c_float, C.c_void_p, C.c_size_t,
                                C.c_void_p, C.c_void_p]),
    "igi_clip_adam_l2": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_float, C.c_double,
                                   C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, C.c_int64, C.c_float,
                                   C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]),
    "igi_clip_adamw": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_float, C.c_double,
                                 C.c_double, C.c_double, C.c_double, C.c_double, C.c_int64, C.c_float, C.c_void_p,
                                 C.c_size_t, C.c_void_p, C.c_void_p]),
    "igi_mlp_grad_floats": (C.c_int64, [C.c_int, C.POINTER(C.c_int32), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "igi_mlp_workspace_bytes": (C.c_size_t, [C.c_int64, C.c_int, C.POINTER(C.c_int32)]),
    "igi_mlp_forward": (C.c_int, [C.c_void_p, C.c_int, C.c_int64, C.c_int, C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                                  C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_int32),
                                  C.c_void_p]),
    "igi_mlp_backward": (C.c_int, [C.c_void_p, C.c_int, C.c_int64, C.c_int, C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                                   C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_void_p, C.c_void_p, C.c_void_p,
                                   C.POINTER(C.c_int32), C.c_void_p, C.c_size_t, C.c_void_p]),
    "igi_linear_workspace_bytes": (C.c_size_t, [C.c_int64, C.c_int, C.c_int]),
    "igi_linear_forward": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int64,
                                     C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "igi_linear_backward": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int,
                                      C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int,
                                      C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]),
    "igi_depth_param_count": (C.c_int64, [C.POINTER(DepthCfg)]),
    "igi_depth_workspace_bytes": (C.c_size_t, [C.POINTER(DepthCfg)]),
    "igi_depth_forward": (C.c_int, [C.POINTER(DepthCfg), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t,
                                    C.c_void_p]),
    "igi_depth_backward": (C.c_int, [C.POINTER(DepthCfg), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.c_size_t, C.c_void_p]),
    "igi_token_param_count": (C.c_int64, [C.POINTER(TokenCfg)]),
    "igi_token_workspace_bytes": (C.c_size_t, [C.POINTER(TokenCfg)]),
    "igi_token_forward": (C.c_int, [C.POINTER(TokenCfg), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t,
                                    C.c_uint64, C.c_void_p]),
    "igi_token_backward": (C.c_int, [C.POINTER(TokenCfg), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.c_size_t, C.c_uint64, C.c_void_p]),
    "igi_tactile_param_count": (C.c_int64, [C.POINTER(TactileCfg)]),
    "igi_tactile_workspace_bytes": (C.c_size_t, [C.POINTER(TactileCfg)]),
    "igi_tactile_forward": (C.c_int, [C.POINTER(TactileCfg), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                      C.c_size_t, C.c_void_p]),
    "igi_tactile_activation_layout": (C.c_int, [C.POINTER(TactileCfg), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "igi_tactile_backward": (C.c_int, [C.POINTER(TactileCfg), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_size_t, C.c_void_p]),
    "igi_spatial_softargmax_forward": (C.c_int, [C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                                 C.c_void_p]),
    "igi_spatial_softargmax_backward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int,
                                                  C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "igi_gather_rows": (C.c_int, [C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_int64), C.POINTER(C.c_void_p), C.c_void_p,
                                  C.c_int64, C.c_int64, C.c_void_p]),
    "igi_cat_cols": (C.c_int, [C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_int64), C.c_void_p, C.c_void_p, C.c_int64,
                               C.c_void_p]),
    "igi_split_cols": (C.c_int, [C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_int64), C.c_void_p, C.c_int64, C.c_void_p]),
    "igi_pointnet_workspace_bytes": (C.c_size_t, [C.c_int64]),
    "igi_pointnet_forward": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_void_p]),
    "igi_pointnet_backward": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_int64,
                                        C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "igi_pointnet_workspace_bytes_multi": (C.c_size_t, [C.c_int64, C.c_int]),
    "igi_pointnet_forward_multi": (C.c_int, [C.c_int, C.c_void_p, C.c_int64, C.c_int64, C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                                             C.POINTER(C.c_void_p), C.c_void_p, C.c_void_p, C.c_void_p]),
    "igi_pointnet_backward_multi": (C.c_int, [C.c_int, C.c_void_p, C.c_int64, C.c_int64, C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                                              C.POINTER(C.c_void_p), C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p,
                                              C.c_size_t, C.c_void_p]),
}

REDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_int)    # igi_reduce_fn(user, bucket, step)

_lib = None


class NativeLibraryError(RuntimeError):
    pass


def exported_symbols():
    """Every symbol include/igi_ppo.h declares."""
    return list(_EXPORTS.keys())


def lib():
    """Load (once) and return the ctypes handle.  Raises NativeLibraryError when the HIP library
    has not been built (python -c 'import __graft_entry__ as g; g.build()')."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NativeLibraryError(
            f"{LIB_PATH} not found: build it with `python -c \"import __graft_entry__ as g; g.build()\"` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    try:
        handle = C.CDLL(LIB_PATH)
    except OSError as e:  # pragma: no cover
        raise NativeLibraryError(f"cannot load {LIB_PATH}: {e}") from e
    for name, (res, args) in _EXPORTS.items():
        try:
            fn = getattr(handle, name)
        except AttributeError as e:
            raise NativeLibraryError(f"{LIB_PATH} does not export {name}") from e
        fn.restype = res
        fn.argtypes = args
    if handle.igi_abi_version() != ABI_VERSION:
        raise NativeLibraryError(
            f"ABI mismatch: library {handle.igi_abi_version()} vs binding {ABI_VERSION}; rebuild")
    _lib = handle
    return _lib


def prof_enable(on):
    lib().igi_prof_enable(1 if on else 0)


def prof_read():
    """[{name, launches, total_ms, flops, bytes}] per kernel class since prof_enable(True)."""
    buf = (ProfEntry * 64)()
    n = lib().igi_prof_read(buf, 64)
    if n < 0:
        check(n, "igi_prof_read")
    return [dict(name=buf[i].name.decode(), launches=int(buf[i].launches), total_ms=float(buf[i].total_ms),
                 flops=float(buf[i].flops), bytes=float(buf[i].bytes)) for i in range(n)]


def check(rc, what=""):
    if rc != 0:
        msg = lib().igi_last_error().decode("utf-8", "replace")
        raise RuntimeError(f"libigi_hip {what} failed (rc={rc}): {msg}")


def ptr(t):
    """Device (or host) address of a torch tensor, None -> NULL."""
    if t is None:
        return None
    return C.c_void_p(t.data_ptr())


def current_stream(device=None):
    import torch
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)
