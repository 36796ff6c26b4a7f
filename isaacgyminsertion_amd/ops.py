"""PyTorch custom ops ``torch.ops.mi355ppo.*`` over the C ABI of libigi_hip.so (include/igi_ppo.h).

This is the op surface SURVEY.md section 8(b) asks for: every native entry point of the learning path is a
dispatcher-registered op with a schema, argument validation that raises ``RuntimeError`` (dtype / device /
contiguity / shape -- the C side only sees raw pointers), a fake (meta) kernel so the ops trace under
FakeTensor / ``torch.compile``, and -- for the student's differentiable blocks -- a registered autograd formula
whose backward is itself an op.  The reference has no native code: each op names the reference lines it replaces.

The implementation of every op is ONE call through ctypes into the ``extern "C"`` function of the same purpose,
made with the tensor's device current and on torch's current stream of that device.  There is no other
implementation: a CPU tensor is refused.

Mutable state (the teacher's flat parameter / gradient / Adam vectors, packed fp64 normalisers, prepared arrays,
workspace) is passed as tensor lists in the field order of ``struct igi_rollout`` / ``struct igi_teacher_state`` and
declared mutated in the schema.
"""
import ctypes as C
from typing import List, Optional, Sequence, Tuple

import torch
from torch.library import register_autograd, register_fake

from . import _lib

NS = "mi355ppo"
Tensor = torch.Tensor
_LIB = torch.library.Library(NS, "DEF")


def _op(schema):
    """Define ``mi355ppo::<name>`` with an explicit schema and register ``fn`` as its (only) kernel.  The low-level
    Library API is used rather than ``torch.library.custom_op``: the same dispatcher registration at ~8 us instead
    of ~40 us of Python per call, which matters for the three ops issued per environment step of a rollout."""
    def deco(fn):
        name = schema.split("(")[0]
        _LIB.define(schema)
        _LIB.impl(name, fn, "CompositeExplicitAutograd")
        return fn
    return deco


def _fake(name):
    return register_fake(f"{NS}::{name}")

ROLLOUT_FIELDS = ("obses", "priv_info", "rewards", "values", "neglogpacs", "dones", "actions", "mus", "sigmas",
                  "last_values")
STATE_FIELDS = ("params", "grads", "adam_m", "adam_v", "rms_obs", "rms_priv", "rms_value", "perm", "returns_raw",
                "advantages", "values_n", "returns_n", "mus_w", "sigmas_w", "stats", "workspace")
_STATE_DTYPES = dict(rms_obs=torch.float64, rms_priv=torch.float64, rms_value=torch.float64, perm=torch.int64,
                     workspace=torch.uint8)


# ---------------------------------------------------------------------------------------------------------------
# validation helpers (the TORCH_CHECKs of the boundary)
# ---------------------------------------------------------------------------------------------------------------
def _check(t, name, dtype=torch.float32, shape=None, dim=None, device=None):
    # fast path: one pass over the properties, no message formatting (three of these ops run per environment step)
    try:
        if t.is_cuda and t.dtype is dtype and t.is_contiguous() and (device is None or t.device == device) \
                and (dim is None or t.dim() == dim):
            if shape is None:
                return t
            ts = t.shape
            if len(ts) == len(shape):
                for s_, d_ in zip(shape, ts):
                    if s_ is not None and s_ != d_:
                        break
                else:
                    return t
    except AttributeError:
        pass
    return _check_slow(t, name, dtype, shape, dim, device)


def _check_rows(t, name, shape=None, device=None):
    """fp32 device tensor (rows, ...) whose rows are dense but may lie any pitch apart (a column slice of a wider tensor,
    read in place by the kernels that take a row pitch) -> (tensor, pitch in floats; 0 = dense)"""
    if t.is_contiguous():
        return _check(t, name, shape=shape, device=device), 0
    if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype is torch.float32 and t.dim() >= 2):
        return _check_slow(t, name, torch.float32, shape, None, device), 0
    if device is not None and t.device != device:
        raise RuntimeError(f"{name}: on {t.device}, expected {device} (all arguments must share one device)")
    row = 1
    for d in range(t.dim() - 1, 0, -1):
        if t.shape[d] != 1 and t.stride(d) != row:
            raise RuntimeError(f"{name}: expected dense rows, got strides {tuple(t.stride())}")
        row *= t.shape[d]
    if t.shape[0] > 1 and t.stride(0) < row:
        raise RuntimeError(f"{name}: overlapping rows, strides {tuple(t.stride())}")
    if shape is not None and (len(shape) != t.dim() or any(s is not None and s != d for s, d in zip(shape, t.shape))):
        raise RuntimeError(f"{name}: expected shape {tuple(shape)}, got {tuple(t.shape)}")
    return t, (t.stride(0) if t.shape[0] > 1 else 0)


def _check_slow(t, name, dtype, shape, dim, device):
    if not isinstance(t, torch.Tensor):
        raise RuntimeError(f"{name}: expected a tensor, got {type(t).__name__}")
    if t.device.type != "cuda":
        raise RuntimeError(f"{name}: expected a HIP (cuda) tensor, got device {t.device} (there is no CPU path)")
    if device is not None and t.device != device:
        raise RuntimeError(f"{name}: on {t.device}, expected {device} (all arguments must share one device)")
    if dtype is not None and t.dtype != dtype:
        raise RuntimeError(f"{name}: expected dtype {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise RuntimeError(f"{name}: expected a contiguous tensor, got strides {tuple(t.stride())}")
    if dim is not None and t.dim() != dim:
        raise RuntimeError(f"{name}: expected {dim} dimensions, got shape {tuple(t.shape)}")
    if shape is not None:
        if len(shape) != t.dim() or any(s is not None and s != d for s, d in zip(shape, t.shape)):
            raise RuntimeError(f"{name}: expected shape {tuple(shape)}, got {tuple(t.shape)}")
    return t


def _stream(t):
    return C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def _p(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _rc(rc, what):
    _lib.check(rc, what)


# ---------------------------------------------------------------------------------------------------------------
# teacher PPO update: cfg / rollout / state marshalling
# ---------------------------------------------------------------------------------------------------------------
def pack_cfg(cfg):
    """struct igi_teacher_cfg -> (int[], float[]) op arguments."""
    icfg = [cfg.obs_dim, cfg.priv_dim, cfg.act_dim, cfg.n_priv_layers] + [int(x) for x in cfg.priv_units] + \
           [cfg.n_layers] + [int(x) for x in cfg.units] + [cfg.num_envs, cfg.horizon, cfg.mini_epochs]
    fcfg = [cfg.gamma, cfg.tau, cfg.lr, cfg.beta1, cfg.beta2, cfg.adam_eps, cfg.e_clip, cfg.critic_coef,
            cfg.entropy_coef, cfg.bounds_loss_coef, cfg.grad_norm, cfg.rms_eps]
    return [int(x) for x in icfg], [float(x) for x in fcfg]


def _unpack_cfg(icfg, fcfg):
    M = _lib.IGI_MAX_LAYERS
    if len(icfg) != 8 + 2 * M or len(fcfg) != 12:
        raise RuntimeError(f"teacher cfg: expected {8 + 2 * M} ints and 12 floats, got {len(icfg)} and {len(fcfg)}")
    c = _lib.TeacherCfg()
    c.obs_dim, c.priv_dim, c.act_dim, c.n_priv_layers = icfg[0:4]
    for i in range(M):
        c.priv_units[i] = icfg[4 + i]
        c.units[i] = icfg[5 + M + i]
    c.n_layers = icfg[4 + M]
    c.num_envs, c.horizon, c.mini_epochs = icfg[5 + 2 * M:8 + 2 * M]
    (c.gamma, c.tau, c.lr, c.beta1, c.beta2, c.adam_eps, c.e_clip, c.critic_coef, c.entropy_coef,
     c.bounds_loss_coef, c.grad_norm, c.rms_eps) = fcfg
    if min(c.obs_dim, c.priv_dim, c.act_dim, c.num_envs, c.horizon, c.mini_epochs) < 1 or \
            not (1 <= c.n_layers <= M) or not (1 <= c.n_priv_layers <= M):
        raise RuntimeError("teacher cfg: non-positive dimension or unsupported layer count")
    return c


_STATE_CACHE = {}


def _teacher_args(state, icfg, fcfg):
    """(cfg struct, state struct, device) for a teacher op call.  The full validation of the sixteen state tensors and
    the two library queries (parameter count, workspace size) run once per distinct (configuration, tensor set); a
    call with the same configuration values and the same tensors (address, size, dtype) reuses the result -- the
    policy-inference op runs once per environment step."""
    key = (tuple(icfg), tuple(fcfg), tuple((t.data_ptr(), t.numel(), t.dtype) for t in state))
    hit = _STATE_CACHE.get(key)
    if hit is not None:
        return hit
    cfg = _unpack_cfg(icfg, fcfg)
    st, dev = _state_struct(state, cfg)
    if len(_STATE_CACHE) > 64:
        _STATE_CACHE.clear()
    _STATE_CACHE[key] = (cfg, st, dev)
    return cfg, st, dev


def _state_struct(state, cfg, need=()):
    """Tensor list in STATE_FIELDS order -> struct igi_teacher_state (validated)."""
    if len(state) != len(STATE_FIELDS):
        raise RuntimeError(f"state: expected {len(STATE_FIELDS)} tensors ({', '.join(STATE_FIELDS)}), got {len(state)}")
    dev = state[0].device
    s = _lib.TeacherState()
    T, N, A = cfg.horizon, cfg.num_envs, cfg.act_dim
    P = int(_lib.lib().igi_teacher_param_count(C.byref(cfg)))
    if P <= 0:
        raise RuntimeError("teacher cfg rejected by the library: " + _lib.lib().igi_last_error().decode())
    shapes = dict(params=(P,), grads=(P,), adam_m=(P,), adam_v=(P,), rms_obs=(2 * cfg.obs_dim + 1,),
                  rms_priv=(2 * cfg.priv_dim + 1,), rms_value=(3,), perm=(T * N,), returns_raw=(T, N, 1),
                  advantages=(T, N), values_n=(T, N, 1), returns_n=(T, N, 1), mus_w=(T, N, A), sigmas_w=(T, N, A))
    for name, t in zip(STATE_FIELDS, state):
        _check(t, f"state.{name}", dtype=_STATE_DTYPES.get(name, torch.float32), shape=shapes.get(name), device=dev)
        setattr(s, name, t.data_ptr())
    if state[14].dim() != 2 or state[14].shape[1] != _lib.IGI_STATS_PER_STEP:
        raise RuntimeError(f"state.stats: expected (steps, {_lib.IGI_STATS_PER_STEP}), got {tuple(state[14].shape)}")
    s.workspace_bytes = state[15].numel()
    need_ws = int(_lib.lib().igi_teacher_workspace_bytes(C.byref(cfg)))
    if s.workspace_bytes < need_ws:
        raise RuntimeError(f"state.workspace: {s.workspace_bytes} bytes, the configuration needs {need_ws}")
    return s, dev


def _rollout_struct(rollout, cfg, dev):
    if len(rollout) != len(ROLLOUT_FIELDS):
        raise RuntimeError(f"rollout: expected {len(ROLLOUT_FIELDS)} tensors ({', '.join(ROLLOUT_FIELDS)}), got {len(rollout)}")
    T, N, A = cfg.horizon, cfg.num_envs, cfg.act_dim
    shapes = dict(obses=(T, N, cfg.obs_dim), priv_info=(T, N, cfg.priv_dim), rewards=(T, N, 1), values=(T, N, 1),
                  neglogpacs=(T, N), dones=(T, N), actions=(T, N, A), mus=(T, N, A), sigmas=(T, N, A),
                  last_values=(N, 1))
    r = _lib.Rollout()
    for name, t in zip(ROLLOUT_FIELDS, rollout):
        _check(t, f"rollout.{name}", dtype=torch.uint8 if name == "dones" else torch.float32, shape=shapes[name],
               device=dev)
        setattr(r, name, t.data_ptr())
    return r


@_op("gae_advnorm(Tensor[] rollout, Tensor(a!)[] state, int[] icfg, float[] fcfg, bool normalize_value) -> ()")
def gae_advnorm(rollout: Sequence[Tensor], state: Sequence[Tensor], icfg: Sequence[int], fcfg: Sequence[float],
                normalize_value: bool) -> None:
    """computer_return + prepare_training + the value-normalisation tail (experience.py:242-263;
    frozen_ppo.py:714-725) + the in-loop normaliser trajectory of the coming update -> igi_teacher_prepare."""
    cfg, st, dev = _teacher_args(state, icfg, fcfg)
    ro = _rollout_struct(rollout, cfg, dev)
    with torch.cuda.device(dev):
        _rc(_lib.lib().igi_teacher_prepare(C.byref(cfg), C.byref(ro), C.byref(st), 1 if normalize_value else 0,
                                           _stream(state[0])), "igi_teacher_prepare")


@_op("ppo_minibatch_fwd_bwd(Tensor[] rollout, Tensor(a!)[] state, int[] icfg, float[] fcfg, int mb_index, int step_slot, int phase) -> ()")
def ppo_minibatch_fwd_bwd(rollout: Sequence[Tensor], state: Sequence[Tensor], icfg: Sequence[int],
                          fcfg: Sequence[float], mb_index: int, step_slot: int, phase: int) -> None:
    """One minibatch: gather + normalise, ActorCriticSplit forward, PPO losses + KL, backward into state.grads
    (experience.py:207-226; models_split.py:166-250; frozen_ppo.py:521-584).  phase -1 = whole step; 0 / 1 = the two
    halves of the data-parallel schedule (trunk bucket final after 0) -> igi_teacher_fwd_bwd[_phase]."""
    cfg, st, dev = _teacher_args(state, icfg, fcfg)
    ro = _rollout_struct(rollout, cfg, dev)
    if phase not in (-1, 0, 1):
        raise RuntimeError(f"phase: expected -1, 0 or 1, got {phase}")
    with torch.cuda.device(dev):
        L = _lib.lib()
        if phase < 0:
            _rc(L.igi_teacher_fwd_bwd(C.byref(cfg), C.byref(ro), C.byref(st), mb_index, step_slot, _stream(state[0])),
                "igi_teacher_fwd_bwd")
        else:
            _rc(L.igi_teacher_fwd_bwd_phase(C.byref(cfg), C.byref(ro), C.byref(st), mb_index, step_slot, phase,
                                            _stream(state[0])), "igi_teacher_fwd_bwd_phase")


@_op("ppo_clip_adam(Tensor(a!)[] state, int[] icfg, float[] fcfg, int step_slot, int adam_t, float grad_scale) -> ()")
def ppo_clip_adam(state: Sequence[Tensor], icfg: Sequence[int], fcfg: Sequence[float], step_slot: int, adam_t: int,
                  grad_scale: float) -> None:
    """param-norm log + clip_grad_norm_ + Adam on the flat vectors, stats row ``step_slot`` (frozen_ppo.py:605-610);
    grad_scale = 1/world after an all-reduce(SUM) -> igi_teacher_apply."""
    cfg, st, dev = _teacher_args(state, icfg, fcfg)
    with torch.cuda.device(dev):
        _rc(_lib.lib().igi_teacher_apply(C.byref(cfg), C.byref(st), step_slot, adam_t, float(grad_scale),
                                         _stream(state[0])), "igi_teacher_apply")


@_op("ppo_update(Tensor[] rollout, Tensor(a!)[] state, int[] icfg, float[] fcfg, int adam_t0) -> ()")
def ppo_update(rollout: Sequence[Tensor], state: Sequence[Tensor], icfg: Sequence[int], fcfg: Sequence[float],
               adam_t0: int) -> None:
    """mini_epochs x n_minibatch optimizer steps enqueued back to back, no host sync (frozen_ppo.py:508-640)
    -> igi_teacher_update."""
    cfg, st, dev = _teacher_args(state, icfg, fcfg)
    ro = _rollout_struct(rollout, cfg, dev)
    with torch.cuda.device(dev):
        _rc(_lib.lib().igi_teacher_update(C.byref(cfg), C.byref(ro), C.byref(st), adam_t0, _stream(state[0])),
            "igi_teacher_update")


_REDUCERS = {}


def register_reducer(fn):
    """fn(bucket, step) -> None: bucket 0 / 1 = start the all-reduce(SUM) of the early / late ranges of state.grads
    (igi_teacher_grad_buckets) without blocking the host; bucket 2 = make the current stream wait for both.  Returns the handle
    ``ppo_update_dp`` takes (ops cannot carry Python callables)."""
    h = max(_REDUCERS, default=0) + 1
    _REDUCERS[h] = fn
    return h


def unregister_reducer(h):
    _REDUCERS.pop(h, None)


@_op("ppo_update_dp(Tensor[] rollout, Tensor(a!)[] state, int[] icfg, float[] fcfg, int adam_t0, float grad_scale, int reducer) -> ()")
def ppo_update_dp(rollout: Sequence[Tensor], state: Sequence[Tensor], icfg: Sequence[int], fcfg: Sequence[float],
                  adam_t0: int, grad_scale: float, reducer: int) -> None:
    """The whole data-parallel update as ONE native call: per optimizer step, trunk backward -> reducer(0) ->
    env_mlp backward -> reducer(1) -> reducer(2) -> clip + Adam with grad_scale = 1/world
    (frozen_ppo.py:508-640, gradient exchange :586-603) -> igi_teacher_update_dp."""
    cfg, st, dev = _teacher_args(state, icfg, fcfg)
    ro = _rollout_struct(rollout, cfg, dev)
    fn = _REDUCERS.get(reducer)
    if fn is None:
        raise RuntimeError(f"reducer: unknown handle {reducer} (see ops.register_reducer)")
    err = []

    def trampoline(user, bucket, step):
        try:
            fn(bucket, step)
            return 0
        except BaseException as e:      # never unwind through the C frame
            err.append(e)
            return 1

    cb = _lib.REDUCE_FN(trampoline)
    with torch.cuda.device(dev):
        rc = _lib.lib().igi_teacher_update_dp(C.byref(cfg), C.byref(ro), C.byref(st), adam_t0, float(grad_scale),
                                              C.cast(cb, C.c_void_p), None, _stream(state[0]))
    if err:
        raise err[0]
    _rc(rc, "igi_teacher_update_dp")


@_op("ppo_update_dp_rccl(Tensor[] rollout, Tensor(a!)[] state, int[] icfg, float[] fcfg, int adam_t0, int comm, bool overlap, Tensor(b!)? stats_sum) -> ()")
def ppo_update_dp_rccl(rollout: Sequence[Tensor], state: Sequence[Tensor], icfg: Sequence[int], fcfg: Sequence[float],
                       adam_t0: int, comm: int, overlap: bool, stats_sum: Optional[Tensor]) -> None:
    """The whole data-parallel update as ONE native call with the gradient exchange issued by the library over its
    own RCCL communicator (``comm`` = igi_comm_t handle from utils.dist.NativeComm): per optimizer step phase 0 ->
    all-reduce of the early bucket on the communication stream -> phase 1 -> all-reduce of the late bucket -> clip +
    Adam with 1/world (frozen_ppo.py:508-640, 586-603) -> igi_teacher_update_dp_rccl."""
    cfg, st, dev = _teacher_args(state, icfg, fcfg)
    ro = _rollout_struct(rollout, cfg, dev)
    if comm == 0:
        raise RuntimeError("comm: null communicator handle")
    if stats_sum is not None:
        _check(stats_sum, "stats_sum", device=dev)
        if stats_sum.numel() != state[STATE_FIELDS.index("stats")].numel():
            raise RuntimeError("stats_sum: must have the shape of state.stats")
    with torch.cuda.device(dev):
        rc = _lib.lib().igi_teacher_update_dp_rccl(C.byref(cfg), C.byref(ro), C.byref(st), adam_t0, C.c_void_p(comm),
                                                   1 if overlap else 0, _p(stats_sum), _stream(state[0]))
    if rc == -6:
        raise RuntimeError("igi_teacher_update_dp_rccl: RCCL: " +
                           _lib.lib().igi_comm_last_error(C.c_void_p(comm)).decode("utf-8", "replace"))
    _rc(rc, "igi_teacher_update_dp_rccl")


@_op("actor_critic_infer(Tensor(a!)[] state, int[] icfg, float[] fcfg, Tensor obs, Tensor priv, bool normalize, bool want_latent) -> (Tensor, Tensor, Tensor)")
def actor_critic_infer(state: Sequence[Tensor], icfg: Sequence[int], fcfg: Sequence[float], obs: Tensor, priv: Tensor,
                       normalize: bool, want_latent: bool) -> Tuple[Tensor, Tensor, Tensor]:
    """model_act / act_inference forward without sampling (models_split.py:120-164; frozen_ppo.py:343-366):
    (mu (rows, act), value (rows, 1) on the normalised scale, latent (rows, priv_units[-1]) or an empty tensor).
    Uses state.workspace as scratch (hence 'mutates') -> igi_teacher_infer."""
    cfg, st, dev = _teacher_args(state, icfg, fcfg)
    _check(obs, "obs", shape=(None, cfg.obs_dim), device=dev)
    _check(priv, "priv", shape=(obs.shape[0], cfg.priv_dim), device=dev)
    rows = obs.shape[0]
    lat_dim = cfg.priv_units[cfg.n_priv_layers - 1]
    mu = torch.empty(rows, cfg.act_dim, dtype=torch.float32, device=dev)
    val = torch.empty(rows, 1, dtype=torch.float32, device=dev)
    lat = torch.empty(rows if want_latent else 0, lat_dim, dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        _rc(_lib.lib().igi_teacher_infer(C.byref(cfg), C.byref(st), _p(obs), _p(priv), rows, 1 if normalize else 0,
                                         _p(mu), _p(val), _p(lat) if want_latent else None, _stream(obs)),
            "igi_teacher_infer")
    return mu, val, lat


@_fake("actor_critic_infer")
def _(state, icfg, fcfg, obs, priv, normalize, want_latent):
    M = _lib.IGI_MAX_LAYERS
    act, lat = icfg[2], icfg[4 + icfg[3] - 1]
    rows = obs.shape[0]
    return (obs.new_empty(rows, act), obs.new_empty(rows, 1), obs.new_empty(rows if want_latent else 0, lat))


# ---------------------------------------------------------------------------------------------------------------
# normaliser, optimizer, rollout bookkeeping, loss
# ---------------------------------------------------------------------------------------------------------------
@_op("rms_update_normalize(Tensor x, Tensor(a!) state, float eps, bool train, bool unnorm) -> Tensor")
def rms_update_normalize(x: Tensor, state: Tensor, eps: float, train: bool, unnorm: bool) -> Tensor:
    """RunningMeanStd.forward (running_mean_std.py:60-93): train -> Chan-merge the batch moments into the packed
    fp64 state [mean(D), var(D), count] first; y = clamp((x-mean)/sqrt(var+eps), +-5), or the ``unnorm`` inverse
    -> igi_rms_forward."""
    _check(x, "x")
    if x.dim() < 1:
        raise RuntimeError("x: expected at least one dimension")
    d = x.shape[-1]
    _check(state, "state", dtype=torch.float64, shape=(2 * d + 1,), device=x.device)
    rows = x.numel() // max(d, 1)
    y = torch.empty_like(x)
    L = _lib.lib()
    with torch.cuda.device(x.device):
        need = L.igi_rms_workspace_bytes(rows, d)
        ws = torch.empty(max(int(need), 16), dtype=torch.uint8, device=x.device)
        _rc(L.igi_rms_forward(_p(x), _p(y), rows, d, _p(state), float(eps), 1 if (train and not unnorm) else 0,
                              1 if unnorm else 0, _p(ws), ws.numel(), _stream(x)), "igi_rms_forward")
    return y


@_fake("rms_update_normalize")
def _(x, state, eps, train, unnorm):
    return torch.empty_like(x)


@_op("clip_adam_step(Tensor(a!) params, Tensor grads, Tensor(b!) exp_avg, Tensor(c!) exp_avg_sq, float max_norm, float lr, float beta1, float beta2, float eps, float weight_decay, float l2, int t, float grad_scale, Tensor(d!) stats) -> ()")
def clip_adam_step(params: Tensor, grads: Tensor, exp_avg: Tensor, exp_avg_sq: Tensor, max_norm: float, lr: float,
                   beta1: float, beta2: float, eps: float, weight_decay: float, l2: float, t: int, grad_scale: float,
                   stats: Tensor) -> None:
    """clip_grad_norm_(max_norm) + Adam / AdamW (decoupled ``weight_decay``) / Adam with coupled ``l2`` on one flat
    fp32 vector; grads are pre-scaled by grad_scale (1/world) (ext_adapt.py:853-855, 1139; runner.py:246-248, 481)
    -> igi_clip_adam_l2."""
    n = _check(params, "params", dim=1).numel()
    for nm, v in (("grads", grads), ("exp_avg", exp_avg), ("exp_avg_sq", exp_avg_sq)):
        _check(v, nm, shape=(n,), device=params.device)
    _check(stats, "stats", shape=(8,), device=params.device)
    if t < 1:
        raise RuntimeError(f"t: Adam's step counter is 1-based, got {t}")
    L = _lib.lib()
    with torch.cuda.device(params.device):
        ws = torch.empty(int(L.igi_clip_adam_workspace_bytes()), dtype=torch.uint8, device=params.device)
        _rc(L.igi_clip_adam_l2(_p(params), _p(grads), _p(exp_avg), _p(exp_avg_sq), n, float(max_norm), float(lr),
                               float(beta1), float(beta2), float(eps), float(weight_decay), float(l2), t,
                               float(grad_scale), _p(ws), ws.numel(), _p(stats), _stream(params)), "igi_clip_adam_l2")


@_op("rollout_act_store(Tensor obs, Tensor priv, Tensor mu, Tensor value_n, Tensor logstd, Tensor noise, Tensor? rms_value, float eps, Tensor(a!) obses_t, Tensor(b!) priv_t, Tensor(c!) actions_t, Tensor(d!) neglogp_t, Tensor(e!) values_t, Tensor(f!) mus_t, Tensor(g!) sigmas_t, Tensor(h!) actions_clamped, Tensor(i!) values_out) -> ()")
def rollout_act_store(obs: Tensor, priv: Tensor, mu: Tensor, value_n: Tensor, logstd: Tensor, noise: Tensor,
                      rms_value: Optional[Tensor], eps: float, obses_t: Tensor, priv_t: Tensor, actions_t: Tensor,
                      neglogp_t: Tensor, values_t: Tensor, mus_t: Tensor, sigmas_t: Tensor, actions_clamped: Tensor,
                      values_out: Tensor) -> None:
    """One environment step of play_steps, policy side (frozen_ppo.py:343-366, 655-665): action = mu + exp(logstd) *
    noise, neglogp, value de-normalisation, seven arena-slot writes, clamp(action, +-1) for env.step
    -> igi_rollout_act_store."""
    dev = obs.device
    n, od = _check(obs, "obs", dim=2).shape
    pd = _check(priv, "priv", shape=(n, None), device=dev).shape[1]
    a = _check(mu, "mu", shape=(n, None), device=dev).shape[1]
    _check(value_n, "value_n", shape=(n, 1), device=dev)
    _check(logstd, "logstd", shape=(a,), device=dev)
    _check(noise, "noise", shape=(n, a), device=dev)
    if rms_value is not None:
        _check(rms_value, "rms_value", dtype=torch.float64, shape=(3,), device=dev)
    for nm, t, shp in (("obses_t", obses_t, (n, od)), ("priv_t", priv_t, (n, pd)), ("actions_t", actions_t, (n, a)),
                       ("mus_t", mus_t, (n, a)), ("sigmas_t", sigmas_t, (n, a)),
                       ("actions_clamped", actions_clamped, (n, a))):
        _check(t, nm, shape=shp, device=dev)
    for nm, t in (("neglogp_t", neglogp_t), ("values_t", values_t), ("values_out", values_out)):
        _check(t, nm, device=dev)
        if t.numel() != n:
            raise RuntimeError(f"{nm}: expected {n} elements, got shape {tuple(t.shape)}")
    with torch.cuda.device(dev):
        _rc(_lib.lib().igi_rollout_act_store(n, od, pd, a, _p(obs), _p(priv), _p(mu), _p(value_n), _p(logstd),
                                             _p(noise), _p(rms_value), float(eps), _p(obses_t), _p(priv_t),
                                             _p(actions_t), _p(neglogp_t), _p(values_t), _p(mus_t), _p(sigmas_t),
                                             _p(actions_clamped), _p(values_out), _stream(obs)),
            "igi_rollout_act_store")


@_op("rollout_policy_step(Tensor(a!)[] state, int[] icfg, float[] fcfg, Tensor obs, Tensor priv, bool normalize, Tensor noise, Tensor? rms_value, Tensor(b!)? obses_t, Tensor(c!)? priv_t, Tensor(d!) actions_t, Tensor(e!) neglogp_t, Tensor(f!) values_t, Tensor(g!) mus_t, Tensor(h!) sigmas_t, Tensor(i!) actions_clamped, Tensor(j!) values_out) -> ()")
def rollout_policy_step(state: Sequence[Tensor], icfg: Sequence[int], fcfg: Sequence[float], obs: Tensor, priv: Tensor,
                        normalize: bool, noise: Tensor, rms_value: Optional[Tensor], obses_t: Optional[Tensor],
                        priv_t: Optional[Tensor], actions_t: Tensor, neglogp_t: Tensor, values_t: Tensor, mus_t: Tensor,
                        sigmas_t: Tensor, actions_clamped: Tensor, values_out: Tensor) -> None:
    """The policy side of one environment step as ONE native call: actor_critic_infer + rollout_act_store fused
    (frozen_ppo.py:343-366, 655-665): normalise, env_mlp, trunk, heads, action = mu + exp(logstd) * noise, neglogp,
    value de-normalisation, arena-slot writes, clamp(action, +-1) -> igi_rollout_policy_step."""
    cfg, st, dev = _teacher_args(state, icfg, fcfg)
    n = _check(obs, "obs", shape=(None, cfg.obs_dim), device=dev).shape[0]
    _check(priv, "priv", shape=(n, cfg.priv_dim), device=dev)
    a = cfg.act_dim
    _check(noise, "noise", shape=(n, a), device=dev)
    if rms_value is not None:
        _check(rms_value, "rms_value", dtype=torch.float64, shape=(3,), device=dev)
    if obses_t is not None:
        _check(obses_t, "obses_t", shape=(n, cfg.obs_dim), device=dev)
    if priv_t is not None:
        _check(priv_t, "priv_t", shape=(n, cfg.priv_dim), device=dev)
    for nm, t in (("actions_t", actions_t), ("mus_t", mus_t), ("sigmas_t", sigmas_t), ("actions_clamped", actions_clamped)):
        _check(t, nm, shape=(n, a), device=dev)
    for nm, t in (("neglogp_t", neglogp_t), ("values_t", values_t), ("values_out", values_out)):
        _check(t, nm, device=dev)
        if t.numel() != n:
            raise RuntimeError(f"{nm}: expected {n} elements, got shape {tuple(t.shape)}")
    with torch.cuda.device(dev):
        _rc(_lib.lib().igi_rollout_policy_step(C.byref(cfg), C.byref(st), _p(obs), _p(priv), n, 1 if normalize else 0,
                                               _p(noise), _p(rms_value), _p(obses_t), _p(priv_t), _p(actions_t),
                                               _p(neglogp_t), _p(values_t), _p(mus_t), _p(sigmas_t),
                                               _p(actions_clamped), _p(values_out), _stream(obs)),
            "igi_rollout_policy_step")


@_op("rollout_env_store(Tensor rewards, Tensor dones, Tensor values, Tensor? time_outs, Tensor? successes, float gamma, bool bootstrap, Tensor(a!) rewards_t, Tensor(b!) dones_t, Tensor(c!) cur_rewards, Tensor(d!) cur_lengths, Tensor(e!) cur_success, Tensor(f!) meter) -> ()")
def rollout_env_store(rewards: Tensor, dones: Tensor, values: Tensor, time_outs: Optional[Tensor],
                      successes: Optional[Tensor], gamma: float, bootstrap: bool, rewards_t: Tensor, dones_t: Tensor,
                      cur_rewards: Tensor, cur_lengths: Tensor, cur_success: Tensor, meter: Tensor) -> None:
    """One environment step of play_steps, env side (frozen_ppo.py:671-700; ext_adapt.py:730-760): dones, shaped
    reward 0.01 r + gamma V time_out, episode accumulators, the windowed meters' sums -> igi_rollout_env_store."""
    dev = rewards.device
    n = _check(rewards, "rewards", dim=1).shape[0]
    _check(dones, "dones", dtype=torch.uint8, shape=(n,), device=dev)
    if time_outs is not None:
        _check(time_outs, "time_outs", dtype=torch.uint8, shape=(n,), device=dev)
    if successes is not None:
        _check(successes, "successes", shape=(n,), device=dev)
    _check(dones_t, "dones_t", dtype=torch.uint8, shape=(n,), device=dev)
    _check(meter, "meter", shape=(4,), device=dev)
    for nm, t in (("values", values), ("rewards_t", rewards_t), ("cur_rewards", cur_rewards),
                  ("cur_lengths", cur_lengths), ("cur_success", cur_success)):
        _check(t, nm, device=dev)
        if t.numel() != n:
            raise RuntimeError(f"{nm}: expected {n} elements, got shape {tuple(t.shape)}")
    with torch.cuda.device(dev):
        _rc(_lib.lib().igi_rollout_env_store(n, _p(rewards), _p(dones), _p(values), _p(time_outs), _p(successes),
                                             float(gamma), 1 if (bootstrap and time_outs is not None) else 0,
                                             _p(rewards_t), _p(dones_t), _p(cur_rewards), _p(cur_lengths),
                                             _p(cur_success), _p(meter), _stream(rewards)), "igi_rollout_env_store")


@_op("bc_loss_fwd_bwd(Tensor mu, Tensor teacher_actions, Tensor weights, bool want_grad) -> (Tensor, Tensor)")
def bc_loss_fwd_bwd(mu: Tensor, teacher_actions: Tensor, weights: Tensor, want_grad: bool) -> Tuple[Tensor, Tensor]:
    """sum(weights * (clamp(mu, +-1) - clamp(teacher, +-1))^2) -- a SUM (ext_adapt.py:812-819) -- and d/dmu in the
    same launch (empty tensor when not wanted) -> igi_bc_loss."""
    rows, act = _check(mu, "mu", dim=2).shape
    _check(teacher_actions, "teacher_actions", shape=(rows, act), device=mu.device)
    _check(weights, "weights", shape=(act,), device=mu.device)
    loss = torch.empty(1, dtype=torch.float32, device=mu.device)
    dmu = torch.empty_like(mu) if want_grad else mu.new_empty(0, act)
    L = _lib.lib()
    with torch.cuda.device(mu.device):
        ws = torch.empty(int(L.igi_bc_loss_workspace_bytes()), dtype=torch.uint8, device=mu.device)
        _rc(L.igi_bc_loss(_p(mu), _p(teacher_actions), _p(weights), rows, act, _p(loss), _p(dmu) if want_grad else None,
                          _p(ws), ws.numel(), _stream(mu)), "igi_bc_loss")
    return loss.reshape(()), dmu


@_fake("bc_loss_fwd_bwd")
def _(mu, teacher_actions, weights, want_grad):
    return mu.new_empty(()), (torch.empty_like(mu) if want_grad else mu.new_empty(0, mu.shape[1]))


@_op("bc_loss(Tensor mu, Tensor teacher_actions, Tensor weights) -> Tensor")
def bc_loss(mu: Tensor, teacher_actions: Tensor, weights: Tensor) -> Tensor:
    """Differentiable (w.r.t. mu) form of bc_loss_fwd_bwd."""
    return torch.ops.mi355ppo.bc_loss_fwd_bwd(mu, teacher_actions, weights, False)[0]


@_fake("bc_loss")
def _(mu, teacher_actions, weights):
    return mu.new_empty(())


def _bc_setup(ctx, inputs, output):
    mu, t, w = inputs
    ctx.save_for_backward(mu, t, w)


def _bc_backward(ctx, g):
    mu, t, w = ctx.saved_tensors
    _, dmu = torch.ops.mi355ppo.bc_loss_fwd_bwd(mu, t, w, True)
    return dmu * g, None, None


register_autograd(f"{NS}::bc_loss", _bc_backward, setup_context=_bc_setup)


@_op("bc_loss_value_grad(Tensor mu, Tensor teacher_actions, Tensor weights) -> (Tensor, Tensor)")
def bc_loss_value_grad(mu: Tensor, teacher_actions: Tensor, weights: Tensor) -> Tuple[Tensor, Tensor]:
    """(loss, d loss / d mu) from ONE igi_bc_loss launch; differentiable w.r.t. mu through the saved second output
    (the training path: ``bc_loss`` above recomputes the gradient with a second launch in its backward)."""
    return torch.ops.mi355ppo.bc_loss_fwd_bwd(mu, teacher_actions, weights, True)


@_fake("bc_loss_value_grad")
def _(mu, teacher_actions, weights):
    return mu.new_empty(()), torch.empty_like(mu)


def _bcvg_setup(ctx, inputs, output):
    ctx.save_for_backward(output[1])
    ctx.mark_non_differentiable(output[1])
    ctx.set_materialize_grads(False)      # (no zero-filled gradient for the second output: one fill launch per step)


def _bcvg_backward(ctx, g, _g_dmu):
    (dmu,) = ctx.saved_tensors
    if g is None:
        return None, None, None
    return dmu * g, None, None


register_autograd(f"{NS}::bc_loss_value_grad", _bcvg_backward, setup_context=_bcvg_setup)


# ---------------------------------------------------------------------------------------------------------------
# GEMM (tests / probes) and nn.Linear with a fused activation
# ---------------------------------------------------------------------------------------------------------------
@_op("gemm_f32(bool a_kcontig, bool b_kcontig, int m, int n, int k, Tensor a, int lda, Tensor b, int ldb, Tensor(a!) c, int ldc, Tensor? bias, Tensor? aux, int ldaux, int epilogue, bool accumulate) -> ()")
def gemm_f32(a_kcontig: bool, b_kcontig: bool, m: int, n: int, k: int, a: Tensor, lda: int, b: Tensor, ldb: int,
             c: Tensor, ldc: int, bias: Optional[Tensor], aux: Optional[Tensor], ldaux: int, epilogue: int,
             accumulate: bool) -> None:
    """C = epilogue(A . B) on the exact-fp32 MFMA kernels with explicit leading dimensions (models_split.py:27-38
    forward / autograd backward products) -> igi_gemm_f32."""
    for nm, t in (("a", a), ("b", b), ("c", c)):
        _check(t, nm, device=a.device)
    for nm, t in (("bias", bias), ("aux", aux)):
        if t is not None:
            _check(t, nm, device=a.device)
    if min(m, n, k) < 0 or min(lda, ldb, ldc) < 1:
        raise RuntimeError("gemm_f32: negative extent or non-positive leading dimension")
    with torch.cuda.device(a.device):
        _rc(_lib.lib().igi_gemm_f32(1 if a_kcontig else 0, 1 if b_kcontig else 0, m, n, k, _p(a), lda, _p(b), ldb,
                                    _p(c), ldc, _p(bias), _p(aux), ldaux, epilogue, 1 if accumulate else 0,
                                    _stream(a)), "igi_gemm_f32")


@_op("linear(Tensor x, Tensor weight, Tensor? bias, int act) -> Tensor")
def linear(x: Tensor, weight: Tensor, bias: Optional[Tensor], act: int) -> Tensor:
    """act(x @ weight.T + bias), act: 0 none / 1 tanh / 2 relu; x (rows, in) with row stride >= in
    (tact.py:137-212, 337-339, 367-369, 407-410) -> igi_linear_forward."""
    if x.device.type != "cuda" or x.dtype != torch.float32 or x.dim() != 2 or (x.shape[1] > 1 and x.stride(1) != 1):
        raise RuntimeError(f"x: expected a 2-D fp32 HIP tensor with unit inner stride, got {x.dtype} {tuple(x.shape)} "
                           f"strides {tuple(x.stride())} on {x.device}")
    rows, in_f = x.shape
    out_f = _check(weight, "weight", shape=(None, in_f), device=x.device).shape[0]
    if bias is not None:
        _check(bias, "bias", shape=(out_f,), device=x.device)
    ldx = x.stride(0) if rows > 1 else in_f
    if ldx < in_f:
        raise RuntimeError(f"x: row stride {ldx} smaller than the row length {in_f}")
    y = torch.empty(rows, out_f, dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        _rc(_lib.lib().igi_linear_forward(_p(x), ldx, _p(weight), _p(bias), _p(y), out_f, rows, in_f, out_f, act,
                                          _stream(x)), "igi_linear_forward")
    return y


@_fake("linear")
def _(x, weight, bias, act):
    return x.new_empty(x.shape[0], weight.shape[0])


@_op("linear_bwd(Tensor x, Tensor weight, Tensor y, Tensor dy, int act, bool need_dx, bool need_dw, bool need_db) -> (Tensor, Tensor, Tensor)")
def linear_bwd(x: Tensor, weight: Tensor, y: Tensor, dy: Tensor, act: int, need_dx: bool, need_dw: bool,
               need_db: bool) -> Tuple[Tensor, Tensor, Tensor]:
    """dz = dy * act'(y); dx = dz . W; dW = dz^T x; db = column sums of dz (fixed-order split sums).  Products that
    are not needed are not computed (their result is an empty tensor) -> igi_linear_backward."""
    rows, in_f = x.shape
    out_f = _check(weight, "weight", shape=(None, in_f), device=x.device).shape[0]
    _check(y, "y", shape=(rows, out_f), device=x.device)
    _check(dy, "dy", shape=(rows, out_f), device=x.device)
    dev = x.device
    ldx = x.stride(0) if rows > 1 else in_f
    dx = torch.empty((rows, in_f) if need_dx else (0, in_f), dtype=torch.float32, device=dev)
    dw = torch.empty((out_f, in_f) if need_dw else (0, in_f), dtype=torch.float32, device=dev)
    db = torch.empty(out_f if (need_db and need_dw) else 0, dtype=torch.float32, device=dev)
    if not (need_dx or need_dw):
        return dx, dw, db
    L = _lib.lib()
    with torch.cuda.device(dev):
        ws = torch.empty(max(int(L.igi_linear_workspace_bytes(rows, in_f, out_f)), 16), dtype=torch.uint8, device=dev)
        _rc(L.igi_linear_backward(_p(x), ldx, _p(weight), _p(y), out_f, _p(dy), out_f, _p(dx) if need_dx else None, in_f,
                                  _p(dw) if need_dw else None, _p(db) if db.numel() else None, rows, in_f, out_f, act,
                                  _p(ws), ws.numel(), _stream(x)), "igi_linear_backward")
    return dx, dw, db


@_fake("linear_bwd")
def _(x, weight, y, dy, act, need_dx, need_dw, need_db):
    rows, in_f = x.shape
    out_f = weight.shape[0]
    return (x.new_empty((rows, in_f) if need_dx else (0, in_f)), x.new_empty((out_f, in_f) if need_dw else (0, in_f)),
            x.new_empty(out_f if (need_db and need_dw) else 0))


def _lin_setup(ctx, inputs, output):
    x, weight, bias, act = inputs
    ctx.save_for_backward(x, weight, output)
    ctx.act, ctx.has_bias = act, bias is not None


def _lin_backward(ctx, dy):
    x, weight, y = ctx.saved_tensors
    need_dx, need_dw = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
    need_db = ctx.has_bias and ctx.needs_input_grad[2]
    dx, dw, db = torch.ops.mi355ppo.linear_bwd(x, weight, y, dy.contiguous(), ctx.act, need_dx, need_dw or need_db,
                                                need_db)
    return (dx if need_dx else None), (dw if need_dw else None), (db if need_db else None), None


register_autograd(f"{NS}::linear", _lin_backward, setup_context=_lin_setup)


@_op("mlp_fwd(Tensor x, Tensor[] weights, Tensor?[] biases, int[] acts) -> Tensor[]")
def mlp_fwd(x: Tensor, weights: List[Tensor], biases: List[Optional[Tensor]], acts: List[int]) -> List[Tensor]:
    """A chain of Linear + activation layers forward in ONE launch (igi_mlp_forward: 32 rows per workgroup through every
    layer, hidden activations in LDS); returns every layer's output (the last one is the chain's; all of them are what
    mlp_bwd reads).  Bit-identical to one ``linear`` call per layer; chains the kernel does not take (a layer wider than
    256 outputs, more than 40 k-chunks of 64, the bf16-input mode: IGI_E_UNSUPPORTED) run as one ``linear`` per layer."""
    n = len(weights)
    if n < 1 or n > 8 or len(biases) != n or len(acts) != n:
        raise RuntimeError("mlp_fwd: 1..8 layers with one weight, bias slot and activation each")
    if x.device.type != "cuda" or x.dtype != torch.float32 or x.dim() != 2 or (x.shape[1] > 1 and x.stride(1) != 1):
        raise RuntimeError("x: expected a 2-D fp32 HIP tensor with unit inner stride")
    rows, in0 = x.shape
    dev = x.device
    dims = [in0]
    for l, (w, b) in enumerate(zip(weights, biases)):
        out_f = _check(w, f"weights[{l}]", shape=(None, dims[-1]), device=dev).shape[0]
        if b is not None:
            _check(b, f"biases[{l}]", shape=(out_f,), device=dev)
        elif acts[l] != 0:
            raise RuntimeError(f"biases[{l}]: an activation needs a bias")
        dims.append(out_f)
    ldx = x.stride(0) if rows > 1 else in0
    if ldx < in0:
        raise RuntimeError(f"x: row stride {ldx} smaller than the row length {in0}")
    ys = [torch.empty(rows, d, dtype=torch.float32, device=dev) for d in dims[1:]]
    if rows == 0:
        return ys
    with torch.cuda.device(dev):
        wp = (C.c_void_p * n)(*[w.data_ptr() for w in weights])
        bp = (C.c_void_p * n)(*[(b.data_ptr() if b is not None else None) for b in biases])
        yp = (C.c_void_p * n)(*[y.data_ptr() for y in ys])
        rc = _lib.lib().igi_mlp_forward(_p(x), ldx, rows, n, (C.c_int32 * (n + 1))(*dims),
                                        (C.c_int32 * n)(*[int(a) for a in acts]), wp, bp, yp, None, _stream(x))
        if rc == _lib.IGI_E_UNSUPPORTED:
            # a chain the one-launch kernel does not take (a layer wider than 256 outputs, more than 40 64-wide k-chunks in
            # all, the opt-in bf16-input mode): igi_mlp_forward's contract is "run the layers one by one" -- same results
            h = x
            for l in range(n):
                h = torch.ops.mi355ppo.linear(h, weights[l], biases[l], int(acts[l]))
                ys[l] = h
            return ys
        _rc(rc, "igi_mlp_forward")
    return ys


@_fake("mlp_fwd")
def _(x, weights, biases, acts):
    return [x.new_empty(x.shape[0], w.shape[0]) for w in weights]


@_op("mlp_bwd(Tensor x, Tensor[] weights, Tensor[] ys, Tensor dy, int[] acts, bool need_dx, bool[] need_w) -> (Tensor, Tensor)")
def mlp_bwd(x: Tensor, weights: List[Tensor], ys: List[Tensor], dy: Tensor, acts: List[int], need_dx: bool,
            need_w: List[bool]) -> Tuple[Tensor, Tensor]:
    """Backward of a chain of Linear + activation layers in one native call (igi_mlp_backward): one grid per layer for
    {weight gradient, data gradient with the lower layer's act' in its epilogue}, one fixed-order sum of every layer's
    row-split partials at the end.  Returns (dx or an empty tensor, the flat gradient buffer [dW_0 | db_0 | dW_1 | ...]
    laid out by igi_mlp_grad_floats; ranges of layers with need_w[l] False are zero)."""
    n = len(weights)
    if n < 1 or n > 8 or len(ys) != n or len(acts) != n or len(need_w) != n:
        raise RuntimeError("mlp_bwd: 1..8 layers with one weight, output, activation and need flag each")
    rows, in0 = x.shape
    dev = x.device
    if x.device.type != "cuda" or x.dtype != torch.float32 or x.dim() != 2 or (in0 > 1 and x.stride(1) != 1):
        raise RuntimeError("x: expected a 2-D fp32 HIP tensor with unit inner stride")
    dims = [in0]
    for l, (w, y) in enumerate(zip(weights, ys)):
        out_f = _check(w, f"weights[{l}]", shape=(None, dims[-1]), device=dev).shape[0]
        _check(y, f"ys[{l}]", shape=(rows, out_f), device=dev)
        dims.append(out_f)
    _check(dy, "dy", shape=(rows, dims[-1]), device=dev)
    L = _lib.lib()
    cd = (C.c_int32 * (n + 1))(*dims)
    nfl = int(L.igi_mlp_grad_floats(n, cd, None, None))
    dx = torch.empty((rows, in0) if need_dx else (0, in0), dtype=torch.float32, device=dev)
    grads = torch.zeros(nfl, dtype=torch.float32, device=dev) if not all(need_w) else \
        torch.empty(nfl, dtype=torch.float32, device=dev)
    if rows == 0:
        return dx, grads.zero_()
    ldx = x.stride(0) if rows > 1 else in0
    with torch.cuda.device(dev):
        nbytes = int(L.igi_mlp_workspace_bytes(rows, n, cd))
        ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=dev)
        wp = (C.c_void_p * n)(*[w.data_ptr() for w in weights])
        yp = (C.c_void_p * n)(*[y.data_ptr() for y in ys])
        _rc(L.igi_mlp_backward(_p(x), ldx, rows, n, cd, (C.c_int32 * n)(*[int(a) for a in acts]), wp, yp, _p(dy),
                               _p(dx) if need_dx else None, _p(grads), (C.c_int32 * n)(*[1 if f else 0 for f in need_w]),
                               _p(ws), ws.numel(), _stream(x)), "igi_mlp_backward")
    return dx, grads


@_fake("mlp_bwd")
def _(x, weights, ys, dy, acts, need_dx, need_w):
    n = sum((w.numel() + 3) // 4 * 4 + (w.shape[0] + 3) // 4 * 4 for w in weights)
    return x.new_empty((x.shape[0], x.shape[1]) if need_dx else (0, x.shape[1])), x.new_empty(n)


_MLP_OFFSETS = {}


def mlp_grad_offsets(dims):
    """(weight offsets, bias offsets, total floats) of the flat gradient buffer mlp_bwd returns for the widths ``dims``
    (cached per width tuple: the chains' backward asks every step)."""
    key = tuple(int(d) for d in dims)
    hit = _MLP_OFFSETS.get(key)
    if hit is None:
        hit = _MLP_OFFSETS[key] = _mlp_grad_offsets(key)
    return hit


def _mlp_grad_offsets(dims):
    n = len(dims) - 1
    cd = (C.c_int32 * (n + 1))(*dims)
    wo, bo = (C.c_int64 * n)(), (C.c_int64 * n)()
    total = int(_lib.lib().igi_mlp_grad_floats(n, cd, wo, bo))
    if total <= 0:
        raise RuntimeError("mlp chain rejected: " + _lib.lib().igi_last_error().decode())
    return list(wo), list(bo), total


# ---------------------------------------------------------------------------------------------------------------
# student encoders: tactile CNN, PointNet, depth backbone, token transformer
# ---------------------------------------------------------------------------------------------------------------
@_op("tactile_cnn_fwd(Tensor x, Tensor params, int latent_dim) -> (Tensor, Tensor)")
def tactile_cnn_fwd(x: Tensor, params: Tensor, latent_dim: int) -> Tuple[Tensor, Tensor]:
    """CNNWithSpatialSoftArgmax forward (tactile_cnn.py:62-79) on (B, 3, H, W), B a multiple of 32: implicit-GEMM
    convolutions + ReLU, spatial soft-argmax, Linear(128, latent).  Returns (y, workspace kept for the backward)
    -> igi_tactile_forward."""
    b, c, h, w = _check(x, "x", dim=4).shape
    if c != 3 or b % 32:
        raise RuntimeError(f"x: expected (32k, 3, H, W), got {tuple(x.shape)}")
    cfg = _lib.TactileCfg(b, h, w, latent_dim)
    L = _lib.lib()
    n = int(L.igi_tactile_param_count(C.byref(cfg)))
    nbytes = int(L.igi_tactile_workspace_bytes(C.byref(cfg)))
    if n <= 0 or nbytes == 0:
        raise RuntimeError("tactile configuration rejected: " + L.igi_last_error().decode())
    _check(params, "params", shape=(n,), device=x.device)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
    y = torch.empty(b, latent_dim, dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        _rc(L.igi_tactile_forward(C.byref(cfg), _p(x), _p(params), _p(y), _p(ws), nbytes, _stream(x)),
            "igi_tactile_forward")
    return y, ws


@_fake("tactile_cnn_fwd")
def _(x, params, latent_dim):
    # workspace sizes are host-side integer arithmetic in the library: exact metadata without a device
    cfg = _lib.TactileCfg(x.shape[0], x.shape[2], x.shape[3], latent_dim)
    nbytes = int(_lib.lib().igi_tactile_workspace_bytes(C.byref(cfg)))
    return x.new_empty(x.shape[0], latent_dim), x.new_empty(nbytes, dtype=torch.uint8)


@_op("tactile_cnn_bwd(Tensor dy, Tensor params, Tensor(a!) ws, int height, int width) -> Tensor")
def tactile_cnn_bwd(dy: Tensor, params: Tensor, ws: Tensor, height: int, width: int) -> Tensor:
    """Gradient of tactile_cnn_fwd w.r.t. the flat parameters (the images get none) -> igi_tactile_backward."""
    b, latent = _check(dy, "dy", dim=2).shape
    cfg = _lib.TactileCfg(b, height, width, latent)
    _check(params, "params", dim=1, device=dy.device)
    _check(ws, "ws", dtype=torch.uint8, device=dy.device)
    grads = torch.empty_like(params)
    with torch.cuda.device(dy.device):
        _rc(_lib.lib().igi_tactile_backward(C.byref(cfg), _p(dy), _p(params), _p(grads), _p(ws), ws.numel(),
                                            _stream(dy)), "igi_tactile_backward")
    return grads


@_fake("tactile_cnn_bwd")
def _(dy, params, ws, height, width):
    return torch.empty_like(params)


def _tac_setup(ctx, inputs, output):
    x, params, latent_dim = inputs
    ctx.save_for_backward(params, output[1])
    ctx.hw = (x.shape[2], x.shape[3])
    ctx.set_materialize_grads(False)


def _tac_backward(ctx, dy, dws):
    params, ws = ctx.saved_tensors
    if dy is None:
        return None, None, None
    return None, torch.ops.mi355ppo.tactile_cnn_bwd(dy.contiguous(), params, ws, ctx.hw[0], ctx.hw[1]), None


register_autograd(f"{NS}::tactile_cnn_fwd", _tac_backward, setup_context=_tac_setup)


@_op("spatial_softargmax_fwd(Tensor x, bool normalize) -> (Tensor, Tensor)")
def spatial_softargmax_fwd(x: Tensor, normalize: bool) -> Tuple[Tensor, Tensor]:
    """SpatialSoftArgmax.forward on (B, C, H, W) (tactile_cnn.py:47-58, coordinate quirk of SURVEY A12 included):
    ((B, 2C) soft-argmax coordinates interleaved (x, y) per channel, (B*C, 2) row statistics for the backward)
    -> igi_spatial_softargmax_forward."""
    b, c, h, w = _check(x, "x", dim=4).shape
    out = torch.empty(b, 2 * c, dtype=torch.float32, device=x.device)
    stat = torch.empty(b * c, 2, dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        _rc(_lib.lib().igi_spatial_softargmax_forward(_p(x), b * c, h, w, 1 if normalize else 0, _p(out), _p(stat),
                                                      _stream(x)), "igi_spatial_softargmax_forward")
    return out, stat


@_fake("spatial_softargmax_fwd")
def _(x, normalize):
    b, c = x.shape[0], x.shape[1]
    return x.new_empty(b, 2 * c), x.new_empty(b * c, 2)


@_op("spatial_softargmax_bwd(Tensor x, Tensor out, Tensor stat, Tensor dout, bool normalize) -> Tensor")
def spatial_softargmax_bwd(x: Tensor, out: Tensor, stat: Tensor, dout: Tensor, normalize: bool) -> Tensor:
    """d/dx of spatial_softargmax_fwd -> igi_spatial_softargmax_backward."""
    b, c, h, w = _check(x, "x", dim=4).shape
    _check(out, "out", shape=(b, 2 * c), device=x.device)
    _check(stat, "stat", shape=(b * c, 2), device=x.device)
    _check(dout, "dout", shape=(b, 2 * c), device=x.device)
    dx = torch.empty_like(x)
    with torch.cuda.device(x.device):
        _rc(_lib.lib().igi_spatial_softargmax_backward(_p(x), _p(out), _p(stat), _p(dout), b * c, h, w,
                                                       1 if normalize else 0, _p(dx), _stream(x)),
            "igi_spatial_softargmax_backward")
    return dx


@_fake("spatial_softargmax_bwd")
def _(x, out, stat, dout, normalize):
    return torch.empty_like(x)


def _ssa_setup(ctx, inputs, output):
    x, normalize = inputs
    ctx.save_for_backward(x, output[0], output[1])
    ctx.normalize = normalize
    ctx.set_materialize_grads(False)


def _ssa_backward(ctx, dout, dstat):
    x, out, stat = ctx.saved_tensors
    if dout is None:
        return None, None
    return torch.ops.mi355ppo.spatial_softargmax_bwd(x, out, stat, dout.contiguous(), ctx.normalize), None


register_autograd(f"{NS}::spatial_softargmax_fwd", _ssa_backward, setup_context=_ssa_setup)


@_op("pointnet_max_fwd(Tensor x, Tensor params) -> (Tensor, Tensor)")
def pointnet_max_fwd(x: Tensor, params: Tensor) -> Tuple[Tensor, Tensor]:
    """PointNet forward (pointnets.py:12-42): Linear(3,64)-GELU-Linear(64,256) per point, max over the points;
    returns (features (B, 256), argmax (B, 256) int32) -> igi_pointnet_forward."""
    x, xpitch = _check_rows(x, "x")
    if x.dim() != 3 or x.shape[2] != 3:
        raise RuntimeError(f"x: expected (B, N, 3) points, got {tuple(x.shape)}")
    b, n, ch = x.shape
    _check(params, "params", shape=(3 * 64 + 64 + 64 * 256 + 256,), device=x.device)
    y = torch.empty(b, 256, dtype=torch.float32, device=x.device)
    idx = torch.empty(b, 256, dtype=torch.int32, device=x.device)
    with torch.cuda.device(x.device):
        _rc(_lib.lib().igi_pointnet_forward(_p(x), xpitch, b, n, _p(params), _p(y), _p(idx), _stream(x)), "igi_pointnet_forward")
    return y, idx


@_fake("pointnet_max_fwd")
def _(x, params):
    return x.new_empty(x.shape[0], 256), x.new_empty(x.shape[0], 256, dtype=torch.int32)


@_op("pointnet_max_bwd(Tensor x, Tensor params, Tensor dy, Tensor idx) -> Tensor")
def pointnet_max_bwd(x: Tensor, params: Tensor, dy: Tensor, idx: Tensor) -> Tensor:
    """Parameter gradient of pointnet_max_fwd: only the <= 256 arg-max points of a sample are revisited
    -> igi_pointnet_backward."""
    x, xpitch = _check_rows(x, "x")
    if x.dim() != 3:
        raise RuntimeError(f"x: expected (B, N, 3) points, got {tuple(x.shape)}")
    b, n, _c = x.shape
    _check(params, "params", dim=1, device=x.device)
    dy, dypitch = _check_rows(dy, "dy", shape=(b, 256), device=x.device)
    _check(idx, "idx", dtype=torch.int32, shape=(b, 256), device=x.device)
    grads = torch.empty_like(params)
    L = _lib.lib()
    with torch.cuda.device(x.device):
        nbytes = int(L.igi_pointnet_workspace_bytes(b))
        ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=x.device)
        _rc(L.igi_pointnet_backward(_p(x), xpitch, b, n, _p(params), _p(dy), dypitch, _p(idx), _p(grads), _p(ws), nbytes, _stream(x)),
            "igi_pointnet_backward")
    return grads


@_fake("pointnet_max_bwd")
def _(x, params, dy, idx):
    return torch.empty_like(params)


def _pn_setup(ctx, inputs, output):
    x, params = inputs
    ctx.save_for_backward(x, params, output[1])
    ctx.set_materialize_grads(False)


def _pn_backward(ctx, dy, didx):
    x, params, idx = ctx.saved_tensors
    if dy is None:
        return None, None
    # a column slice of the concatenated encodings' gradient is read in place; anything else (an expanded scalar ...) is copied
    if not (dy.dim() == 2 and dy.stride(1) == 1 and (dy.shape[0] == 1 or dy.stride(0) >= dy.shape[1])):
        dy = dy.contiguous()
    return None, torch.ops.mi355ppo.pointnet_max_bwd(x, params, dy, idx)


register_autograd(f"{NS}::pointnet_max_fwd", _pn_backward, setup_context=_pn_setup)


def _pn_multi_args(x, params, points, who):
    """checks of the multi-object PointNet ops -> (x, xpitch, batch, nobj, x_off[], npoints[], params pointer array)"""
    x, xpitch = _check_rows(x, "x")
    if x.dim() != 3 or x.shape[2] != 3:
        raise RuntimeError(f"x: expected (B, N, 3) points, got {tuple(x.shape)}")
    nobj = len(params)
    if nobj < 1 or nobj > 4 or len(points) != nobj:
        raise RuntimeError(f"{who}: 1..4 objects with one parameter vector and one point count each")
    b, n, _c = x.shape
    if any(int(q) < 1 for q in points) or sum(int(q) for q in points) > n:
        raise RuntimeError(f"points: {list(points)} do not fit the {n} points of a cloud")
    for i, p in enumerate(params):
        _check(p, f"params[{i}]", shape=(3 * 64 + 64 + 64 * 256 + 256,), device=x.device)
    offs, o = [], 0
    for q in points:
        offs.append(3 * o)
        o += int(q)
    if xpitch == 0:
        xpitch = 3 * n
    return (x, xpitch, b, nobj, (C.c_int32 * nobj)(*offs), (C.c_int32 * nobj)(*[int(q) for q in points]),
            (C.c_void_p * nobj)(*[p.data_ptr() for p in params]))


@_op("pointnet_max_fwd_multi(Tensor x, Tensor[] params, int[] points) -> (Tensor, Tensor)")
def pointnet_max_fwd_multi(x: Tensor, params: List[Tensor], points: List[int]) -> Tuple[Tensor, Tensor]:
    """Several PointNets over consecutive slices of ONE cloud tensor in one launch (tact.py:542-571: plug = obs_pcl[:, :400],
    socket = obs_pcl[:, 400:800], each with its own weights; the encodings concatenated): object i encodes points[i] points
    starting where object i - 1 ended.  Returns (features (B, objects * 256), argmax (B, objects * 256) int32)
    -> igi_pointnet_forward_multi.  Per object bit-identical to pointnet_max_fwd on the slice."""
    x, xpitch, b, nobj, offs, npts, pp = _pn_multi_args(x, params, points, "pointnet_max_fwd_multi")
    y = torch.empty(b, nobj * 256, dtype=torch.float32, device=x.device)
    idx = torch.empty(b, nobj * 256, dtype=torch.int32, device=x.device)
    with torch.cuda.device(x.device):
        _rc(_lib.lib().igi_pointnet_forward_multi(nobj, _p(x), xpitch, b, offs, npts, pp, _p(y), _p(idx), _stream(x)),
            "igi_pointnet_forward_multi")
    return y, idx


@_fake("pointnet_max_fwd_multi")
def _(x, params, points):
    return x.new_empty(x.shape[0], 256 * len(params)), x.new_empty(x.shape[0], 256 * len(params), dtype=torch.int32)


@_op("pointnet_max_bwd_multi(Tensor x, Tensor[] params, int[] points, Tensor dy, Tensor idx) -> Tensor")
def pointnet_max_bwd_multi(x: Tensor, params: List[Tensor], points: List[int], dy: Tensor, idx: Tensor) -> Tensor:
    """Parameter gradients of pointnet_max_fwd_multi, (objects, 16896): one launch for every object + one fixed-order sum of
    the per-workgroup records -> igi_pointnet_backward_multi."""
    x, xpitch, b, nobj, offs, npts, pp = _pn_multi_args(x, params, points, "pointnet_max_bwd_multi")
    dy, dypitch = _check_rows(dy, "dy", shape=(b, nobj * 256), device=x.device)
    _check(idx, "idx", dtype=torch.int32, shape=(b, nobj * 256), device=x.device)
    grads = torch.empty(nobj, params[0].numel(), dtype=torch.float32, device=x.device)
    L = _lib.lib()
    with torch.cuda.device(x.device):
        nbytes = int(L.igi_pointnet_workspace_bytes_multi(b, nobj))
        ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=x.device)
        _rc(L.igi_pointnet_backward_multi(nobj, _p(x), xpitch, b, offs, npts, pp, _p(dy), dypitch, _p(idx), _p(grads), _p(ws),
                                          nbytes, _stream(x)), "igi_pointnet_backward_multi")
    return grads


@_fake("pointnet_max_bwd_multi")
def _(x, params, points, dy, idx):
    return x.new_empty(len(params), params[0].numel())


def _pnm_setup(ctx, inputs, output):
    x, params, points = inputs
    ctx.save_for_backward(x, output[1], *params)
    ctx.points = [int(q) for q in points]
    ctx.set_materialize_grads(False)


def _pnm_backward(ctx, dy, didx):
    x, idx = ctx.saved_tensors[:2]
    params = list(ctx.saved_tensors[2:])
    if dy is None:
        return None, None, None
    if not (dy.dim() == 2 and dy.stride(1) == 1 and (dy.shape[0] == 1 or dy.stride(0) >= dy.shape[1])):
        dy = dy.contiguous()
    g = torch.ops.mi355ppo.pointnet_max_bwd_multi(x, params, ctx.points, dy, idx)
    return None, [g[i] for i in range(len(params))], None


register_autograd(f"{NS}::pointnet_max_fwd_multi", _pnm_backward, setup_context=_pnm_setup)


@_op("depth_backbone_fwd(Tensor x, Tensor params, int latent_dim) -> (Tensor, Tensor)")
def depth_backbone_fwd(x: Tensor, params: Tensor, latent_dim: int) -> Tuple[Tensor, Tensor]:
    """DepthOnlyFCBackbone54x96 forward (tact.py:81-113) on (32k, 1, 54, 96) -> igi_depth_forward."""
    if x.dim() != 4 or tuple(x.shape[1:]) != (1, 54, 96) or x.shape[0] % 32:
        raise RuntimeError(f"x: expected (32k, 1, 54, 96) images, got {tuple(x.shape)}")
    _check(x, "x")
    b = x.shape[0]
    cfg = _lib.DepthCfg(b, latent_dim)
    L = _lib.lib()
    _check(params, "params", shape=(int(L.igi_depth_param_count(C.byref(cfg))),), device=x.device)
    nbytes = int(L.igi_depth_workspace_bytes(C.byref(cfg)))
    ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
    y = torch.empty(b, latent_dim, dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        _rc(L.igi_depth_forward(C.byref(cfg), _p(x), _p(params), _p(y), _p(ws), nbytes, _stream(x)), "igi_depth_forward")
    return y, ws


@_fake("depth_backbone_fwd")
def _(x, params, latent_dim):
    nbytes = int(_lib.lib().igi_depth_workspace_bytes(C.byref(_lib.DepthCfg(x.shape[0], latent_dim))))
    return x.new_empty(x.shape[0], latent_dim), x.new_empty(nbytes, dtype=torch.uint8)


@_op("depth_backbone_bwd(Tensor x, Tensor dy, Tensor params, Tensor(a!) ws) -> Tensor")
def depth_backbone_bwd(x: Tensor, dy: Tensor, params: Tensor, ws: Tensor) -> Tensor:
    """Parameter gradient of depth_backbone_fwd -> igi_depth_backward."""
    b, latent = _check(dy, "dy", dim=2).shape
    _check(x, "x", shape=(b, 1, 54, 96), device=dy.device)
    _check(params, "params", dim=1, device=dy.device)
    _check(ws, "ws", dtype=torch.uint8, device=dy.device)
    cfg = _lib.DepthCfg(b, latent)
    grads = torch.empty_like(params)
    with torch.cuda.device(dy.device):
        _rc(_lib.lib().igi_depth_backward(C.byref(cfg), _p(x), _p(dy), _p(params), _p(grads), _p(ws), ws.numel(),
                                          _stream(dy)), "igi_depth_backward")
    return grads


@_fake("depth_backbone_bwd")
def _(x, dy, params, ws):
    return torch.empty_like(params)


def _dep_setup(ctx, inputs, output):
    x, params, latent_dim = inputs
    ctx.save_for_backward(x, params, output[1])
    ctx.set_materialize_grads(False)


def _dep_backward(ctx, dy, dws):
    x, params, ws = ctx.saved_tensors
    if dy is None:
        return None, None, None
    return None, torch.ops.mi355ppo.depth_backbone_bwd(x, dy.contiguous(), params, ws), None


register_autograd(f"{NS}::depth_backbone_fwd", _dep_backward, setup_context=_dep_setup)


def _token_cfg(x, nhead, ff, layers, dropout, training):
    B, S, d = x.shape
    return _lib.TokenCfg(B, S, d, nhead, ff, layers, float(dropout), 1 if training else 0)


@_op("token_encoder_fwd(Tensor x, Tensor params, int nhead, int ff, int layers, float dropout, bool training, int seed) -> (Tensor, Tensor)")
def token_encoder_fwd(x: Tensor, params: Tensor, nhead: int, ff: int, layers: int, dropout: float, training: bool,
                      seed: int) -> Tuple[Tensor, Tensor]:
    """``layers`` x TransformerEncoderLayer(d, nhead, ff, gelu, batch_first, norm_first) over (B, S <= 8, d)
    (tact.py:143-148); dropout masks from a counter hash of ``seed`` -> igi_token_forward."""
    _check(x, "x", dim=3)
    cfg = _token_cfg(x, nhead, ff, layers, dropout, training)
    L = _lib.lib()
    n = int(L.igi_token_param_count(C.byref(cfg)))
    if n < 0:
        _rc(n, "igi_token_param_count")
    _check(params, "params", shape=(n,), device=x.device)
    y = torch.empty_like(x)
    nbytes = int(L.igi_token_workspace_bytes(C.byref(cfg)))
    ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
    with torch.cuda.device(x.device):
        _rc(L.igi_token_forward(C.byref(cfg), _p(x), _p(params), _p(y), _p(ws), nbytes, C.c_uint64(seed), _stream(x)),
            "igi_token_forward")
    return y, ws


@_fake("token_encoder_fwd")
def _(x, params, nhead, ff, layers, dropout, training, seed):
    nbytes = int(_lib.lib().igi_token_workspace_bytes(C.byref(_token_cfg(x, nhead, ff, layers, dropout, training))))
    return torch.empty_like(x), x.new_empty(nbytes, dtype=torch.uint8)


@_op("token_encoder_bwd(Tensor dy, Tensor params, Tensor(a!) ws, int nhead, int ff, int layers, float dropout, bool training, int seed) -> (Tensor, Tensor)")
def token_encoder_bwd(dy: Tensor, params: Tensor, ws: Tensor, nhead: int, ff: int, layers: int, dropout: float,
                      training: bool, seed: int) -> Tuple[Tensor, Tensor]:
    """(dx, parameter gradient) of token_encoder_fwd; the dropout masks are regenerated -> igi_token_backward."""
    _check(dy, "dy", dim=3)
    _check(params, "params", dim=1, device=dy.device)
    _check(ws, "ws", dtype=torch.uint8, device=dy.device)
    cfg = _token_cfg(dy, nhead, ff, layers, dropout, training)
    dx = torch.empty_like(dy)
    grads = torch.empty_like(params)
    with torch.cuda.device(dy.device):
        _rc(_lib.lib().igi_token_backward(C.byref(cfg), _p(dy), _p(params), _p(dx), _p(grads), _p(ws), ws.numel(),
                                          C.c_uint64(seed), _stream(dy)), "igi_token_backward")
    return dx, grads


@_fake("token_encoder_bwd")
def _(dy, params, ws, nhead, ff, layers, dropout, training, seed):
    return torch.empty_like(dy), torch.empty_like(params)


def _tok_setup(ctx, inputs, output):
    x, params, nhead, ff, layers, dropout, training, seed = inputs
    ctx.save_for_backward(params, output[1])
    ctx.args = (nhead, ff, layers, dropout, training, seed)
    ctx.set_materialize_grads(False)


def _tok_backward(ctx, dy, dws):
    params, ws = ctx.saved_tensors
    if dy is None:
        return (None,) * 8
    dx, grads = torch.ops.mi355ppo.token_encoder_bwd(dy.contiguous(), params, ws, *ctx.args)
    return (dx, grads) + (None,) * 6


register_autograd(f"{NS}::token_encoder_fwd", _tok_backward, setup_context=_tok_setup)

def _ptr_array(ts):
    return (C.c_void_p * len(ts))(*[t.data_ptr() for t in ts])


@_op("gather_rows(Tensor[] arenas, Tensor rows) -> Tensor[]")
def gather_rows(arenas: List[Tensor], rows: Tensor) -> List[Tensor]:
    """``[a.index_select(0, rows) for a in arenas]`` in ONE launch (igi_gather_rows): the minibatch of every key a student
    step reads (experience.py:117-139).  arenas: fp32 (rows_total, ...) contiguous, all with the same leading dimension."""
    n = len(arenas)
    if n < 1 or n > 8:
        raise RuntimeError("gather_rows: 1..8 arenas")
    dev = arenas[0].device
    total = arenas[0].shape[0]
    _check(rows, "rows", dtype=torch.int64, dim=1, device=dev)
    for k, a in enumerate(arenas):
        _check(a, f"arenas[{k}]", device=dev)
        if a.dim() < 1 or a.shape[0] != total or a.numel() == 0:
            raise RuntimeError(f"arenas[{k}]: expected ({total}, ...) non-empty, got {tuple(a.shape)}")
    nr = rows.numel()
    outs = [torch.empty((nr,) + tuple(a.shape[1:]), dtype=torch.float32, device=dev) for a in arenas]
    if nr == 0:
        return outs
    width = (C.c_int64 * n)(*[a.numel() // total for a in arenas])
    with torch.cuda.device(dev):
        _rc(_lib.lib().igi_gather_rows(n, _ptr_array(arenas), width, _ptr_array(outs), _p(rows), nr, total, _stream(rows)),
            "igi_gather_rows")
    return outs


@_fake("gather_rows")
def _(arenas, rows):
    return [a.new_empty((rows.numel(),) + tuple(a.shape[1:])) for a in arenas]


@_op("cat_cols(Tensor[] parts, Tensor? add) -> Tensor")
def cat_cols(parts: List[Tensor], add: Optional[Tensor]) -> Tensor:
    """``torch.cat(parts, dim=1) (+ add)`` for 2-D fp32 parts with the same number of rows, ``add`` one row of the
    concatenated width (the positional encoding) -> igi_cat_cols.  Autograd: the gradient is split back by ONE launch
    (igi_split_cols), every part's gradient dense."""
    n = len(parts)
    if n < 1 or n > 8:
        raise RuntimeError("cat_cols: 1..8 parts")
    dev = parts[0].device
    rows = parts[0].shape[0] if parts[0].dim() == 2 else -1
    for k, p in enumerate(parts):
        _check(p, f"parts[{k}]", shape=(rows, None), device=dev)
    total = sum(p.shape[1] for p in parts)
    if add is not None:
        _check(add, "add", shape=(total,), device=dev)
    out = torch.empty(rows, total, dtype=torch.float32, device=dev)
    if rows == 0 or total == 0:
        return out
    if any(p.shape[1] == 0 for p in parts):
        raise RuntimeError("cat_cols: empty part")
    width = (C.c_int64 * n)(*[p.shape[1] for p in parts])
    with torch.cuda.device(dev):
        _rc(_lib.lib().igi_cat_cols(n, _ptr_array(parts), width, _p(out), _p(add), rows, _stream(out)), "igi_cat_cols")
    return out


@_fake("cat_cols")
def _(parts, add):
    return parts[0].new_empty(parts[0].shape[0], sum(p.shape[1] for p in parts))


@_op("split_cols(Tensor cat, int[] widths) -> Tensor[]")
def split_cols(cat: Tensor, widths: List[int]) -> List[Tensor]:
    """``[c.contiguous() for c in cat.split(widths, dim=1)]`` in ONE launch -> igi_split_cols."""
    n = len(widths)
    if n < 1 or n > 8 or any(w < 1 for w in widths):
        raise RuntimeError("split_cols: 1..8 positive widths")
    _check(cat, "cat", shape=(None, sum(widths)))
    rows = cat.shape[0]
    outs = [torch.empty(rows, w, dtype=torch.float32, device=cat.device) for w in widths]
    if rows == 0:
        return outs
    width = (C.c_int64 * n)(*widths)
    with torch.cuda.device(cat.device):
        _rc(_lib.lib().igi_split_cols(n, _ptr_array(outs), width, _p(cat), rows, _stream(cat)), "igi_split_cols")
    return outs


@_fake("split_cols")
def _(cat, widths):
    return [cat.new_empty(cat.shape[0], w) for w in widths]


def _cat_setup(ctx, inputs, output):
    parts, _add = inputs
    ctx.widths = [p.shape[1] for p in parts]
    ctx.has_add = _add is not None


def _cat_backward(ctx, dy):
    return list(torch.ops.mi355ppo.split_cols(dy.contiguous(), ctx.widths)), (dy.sum(0) if ctx.has_add and ctx.needs_input_grad[1] else None)


register_autograd(f"{NS}::cat_cols", _cat_backward, setup_context=_cat_setup)

# ops that only mutate their arguments: the fake kernel returns nothing
for _n in ("gae_advnorm", "ppo_minibatch_fwd_bwd", "ppo_clip_adam", "ppo_update", "ppo_update_dp", "ppo_update_dp_rccl",
           "clip_adam_step",
           "rollout_act_store", "rollout_policy_step", "rollout_env_store", "gemm_f32"):
    register_fake(f"{NS}::{_n}")(lambda *a, **k: None)

OP_NAMES = ["gae_advnorm", "ppo_minibatch_fwd_bwd", "ppo_clip_adam", "ppo_update", "ppo_update_dp", "ppo_update_dp_rccl",
            "actor_critic_infer", "rms_update_normalize", "clip_adam_step", "rollout_act_store", "rollout_policy_step",
            "rollout_env_store",
            "bc_loss_fwd_bwd", "bc_loss", "bc_loss_value_grad", "gemm_f32", "linear", "linear_bwd", "mlp_fwd", "mlp_bwd", "tactile_cnn_fwd", "tactile_cnn_bwd", "spatial_softargmax_fwd", "spatial_softargmax_bwd",
            "pointnet_max_fwd", "pointnet_max_bwd", "pointnet_max_fwd_multi", "pointnet_max_bwd_multi", "depth_backbone_fwd", "depth_backbone_bwd", "token_encoder_fwd",
            "token_encoder_bwd", "gather_rows", "cat_cols", "split_cols"]
