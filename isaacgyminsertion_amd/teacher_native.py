"""Device-side state of the teacher PPO update and its calls into libigi_hip.so.

``TeacherEngine`` owns the HBM-resident arenas (flat parameters / gradients / Adam moments, packed
fp64 normaliser states, prepared per-update arrays, workspace) and exposes the C-ABI entry points
as methods.  The reference-shaped classes (algo.ppo.frozen_ppo.PPO, ExperienceBuffer,
ActorCriticSplit, RunningMeanStd) hold *views* into these tensors.

torch is used here for device memory, streams and (multi-GPU) torch.distributed only.
"""
import ctypes as C
import os
from collections import OrderedDict

import torch

from . import _lib, ops

# hyper-parameter defaults: cfg/train/FactoryTaskInsertionTactilePPOv2.yaml:28-45, Adam defaults
DEFAULT_HP = dict(gamma=0.99, tau=0.95, lr=2.5e-4, beta1=0.9, beta2=0.999, adam_eps=1e-8, e_clip=0.2,
                  critic_coef=4.0, entropy_coef=0.0, bounds_loss_coef=1e-4, grad_norm=1.0,
                  truncate_grads=True, rms_eps=1e-5, normalize_value=True)

ROLLOUT_KEYS = ("obses", "priv_info", "rewards", "values", "neglogpacs", "dones", "actions", "mus",
                "sigmas", "last_values")


def teacher_param_names(n_priv_layers, n_layers):
    """ActorCriticSplit.state_dict() key order (models_split.py:73-106)."""
    names = ["sigma"]
    for i in range(n_priv_layers):
        names += [f"env_mlp.mlp.{2 * i}.weight", f"env_mlp.mlp.{2 * i}.bias"]
    for net in ("actor_mlp", "critic_mlp"):
        for i in range(n_layers):
            names += [f"{net}.mlp.{2 * i}.weight", f"{net}.mlp.{2 * i}.bias"]
    names += ["value.weight", "value.bias", "mu.weight", "mu.bias"]
    return names


def teacher_param_shapes(obs_dim, priv_dim, act_dim, units, priv_units):
    shapes = OrderedDict()
    shapes["sigma"] = (act_dim,)
    d = priv_dim
    for i, u in enumerate(priv_units):
        shapes[f"env_mlp.mlp.{2 * i}.weight"] = (u, d)
        shapes[f"env_mlp.mlp.{2 * i}.bias"] = (u,)
        d = u
    for net in ("actor_mlp", "critic_mlp"):
        d = obs_dim + priv_units[-1]
        for i, u in enumerate(units):
            shapes[f"{net}.mlp.{2 * i}.weight"] = (u, d)
            shapes[f"{net}.mlp.{2 * i}.bias"] = (u,)
            d = u
    shapes["value.weight"] = (1, units[-1])
    shapes["value.bias"] = (1,)
    shapes["mu.weight"] = (act_dim, units[-1])
    shapes["mu.bias"] = (act_dim,)
    return shapes


def make_cfg(obs_dim, priv_dim, act_dim, units, priv_units, num_envs, horizon, mini_epochs, **hp):
    h = dict(DEFAULT_HP)
    h.update(hp)
    if len(units) > _lib.IGI_MAX_LAYERS or len(priv_units) > _lib.IGI_MAX_LAYERS:
        raise ValueError(f"at most {_lib.IGI_MAX_LAYERS} layers per MLP are supported")
    c = _lib.TeacherCfg()
    c.obs_dim, c.priv_dim, c.act_dim = obs_dim, priv_dim, act_dim
    c.n_priv_layers, c.n_layers = len(priv_units), len(units)
    for i, u in enumerate(priv_units):
        c.priv_units[i] = int(u)
    for i, u in enumerate(units):
        c.units[i] = int(u)
    c.num_envs, c.horizon, c.mini_epochs = num_envs, horizon, mini_epochs
    c.gamma, c.tau = float(h["gamma"]), float(h["tau"])
    c.lr, c.beta1, c.beta2, c.adam_eps = float(h["lr"]), float(h["beta1"]), float(h["beta2"]), float(h["adam_eps"])
    c.e_clip, c.critic_coef = float(h["e_clip"]), float(h["critic_coef"])
    c.entropy_coef, c.bounds_loss_coef = float(h["entropy_coef"]), float(h["bounds_loss_coef"])
    c.grad_norm = float(h["grad_norm"]) if h["truncate_grads"] else 0.0
    c.rms_eps = float(h["rms_eps"])
    return c, h


def param_layout(cfg):
    """(padded length, [(offset, size)] per tensor in state_dict order) from the library."""
    L = _lib.lib()
    n = L.igi_teacher_param_offsets(C.byref(cfg), None, None, 0)
    if n < 0:
        _lib.check(n, "igi_teacher_param_offsets")
    off = (C.c_int64 * n)()
    sz = (C.c_int64 * n)()
    L.igi_teacher_param_offsets(C.byref(cfg), off, sz, n)
    total = L.igi_teacher_param_count(C.byref(cfg))
    return int(total), [(int(off[i]), int(sz[i])) for i in range(n)]


class TeacherEngine:
    def __init__(self, num_envs, horizon, mini_epochs, units=(512, 256, 128), priv_units=(256, 128, 8),
                 obs_dim=15, priv_dim=64, act_dim=6, device="cuda:0", perm=None, **hp):
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("TeacherEngine needs a HIP device (there is no CPU path)")
        self.L = _lib.lib()
        self.cfg, self.hp = make_cfg(obs_dim, priv_dim, act_dim, units, priv_units, num_envs, horizon,
                                     mini_epochs, **hp)
        self.N, self.T, self.E = num_envs, horizon, mini_epochs
        self.B = num_envs * horizon
        self.mb = self.B // mini_epochs
        self.n_mb = self.B // self.mb
        self.obs_dim, self.priv_dim, self.act_dim = obs_dim, priv_dim, act_dim
        self.units, self.priv_units = list(units), list(priv_units)
        self.shapes = teacher_param_shapes(obs_dim, priv_dim, act_dim, self.units, self.priv_units)
        self.P, self.layout = param_layout(self.cfg)
        assert len(self.layout) == len(self.shapes)
        dev = self.device
        f32 = dict(dtype=torch.float32, device=dev)
        self.params = torch.zeros(self.P, **f32)
        self.grads = torch.zeros(self.P, **f32)
        self.adam_m = torch.zeros(self.P, **f32)
        self.adam_v = torch.zeros(self.P, **f32)
        self.adam_t = 0
        self.rms_obs = self._fresh_rms(obs_dim)
        self.rms_priv = self._fresh_rms(priv_dim)
        self.rms_value = self._fresh_rms(1)
        T, N, A = horizon, num_envs, act_dim
        self.returns_raw = torch.zeros(T, N, 1, **f32)
        self.advantages = torch.zeros(T, N, **f32)
        self.values_n = torch.zeros(T, N, 1, **f32)
        self.returns_n = torch.zeros(T, N, 1, **f32)
        self.mus_w = torch.zeros(T, N, A, **f32)
        self.sigmas_w = torch.zeros(T, N, A, **f32)
        self.stats = torch.zeros(mini_epochs * self.n_mb, _lib.IGI_STATS_PER_STEP, **f32)
        wbytes = self.L.igi_teacher_workspace_bytes(C.byref(self.cfg))
        if wbytes == 0:
            raise RuntimeError("igi_teacher_workspace_bytes rejected the configuration: "
                               + self.L.igi_last_error().decode())
        self.workspace = torch.zeros(wbytes, dtype=torch.uint8, device=dev)
        self.workspace_trial_ms = None      # see tune_workspace
        if perm is None:
            perm = torch.randperm(self.B, device=dev)          # experience.py:202, drawn once
        self.perm = perm.to(device=dev, dtype=torch.int64).contiguous()
        self._ro = None

    def _fresh_rms(self, d):
        s = torch.zeros(2 * d + 1, dtype=torch.float64, device=self.device)
        s[d:2 * d] = 1.0   # running_var = 1 (running_mean_std.py:45)
        s[2 * d] = 1.0     # count = 1     (running_mean_std.py:46)
        return s

    # ---- parameters --------------------------------------------------------------------------
    def param_views(self, flat=None):
        flat = self.params if flat is None else flat
        out = OrderedDict()
        for (name, shape), (off, size) in zip(self.shapes.items(), self.layout):
            out[name] = flat[off:off + size].view(shape)
        return out

    def load_params(self, state_dict):
        views = self.param_views()
        for k, v in views.items():
            v.copy_(state_dict[k].to(self.device, torch.float32))

    def packed(self, flat=None):
        """Unpadded concatenation in state_dict order (what the reference's torch.cat produces)."""
        return torch.cat([v.reshape(-1) for v in self.param_views(flat).values()])

    # ---- native calls (torch.ops.mi355ppo.*: schema-checked, device-guarded; see ops.py) -------------------------
    def state_list(self):
        """The sixteen tensors of struct igi_teacher_state, in field order."""
        return [getattr(self, k) for k in ops.STATE_FIELDS]

    def _cfg_args(self):
        """The packed igi_teacher_cfg, rebuilt from the struct on every call: trainers mutate ``cfg.lr`` and
        ``ExperienceBuffer.computer_return(last_values, gamma, tau)`` mutates ``cfg.gamma`` / ``cfg.tau``
        (experience.py:242), so no field may be cached.  32 scalars -- noise next to one native call."""
        return ops.pack_cfg(self.cfg)

    def set_rollout(self, ro):
        """ro: dict of time-major device tensors (ROLLOUT_KEYS); kept referenced, not copied."""
        keep = []
        for k in ROLLOUT_KEYS:
            t = ro[k]
            want = torch.uint8 if k == "dones" else torch.float32
            if t.device != self.device or t.dtype != want or not t.is_contiguous():
                t = t.to(device=self.device, dtype=want).contiguous()
            keep.append(t)
        self._ro = keep

    def tune_workspace(self, trials=None):
        """Pick the workspace ALLOCATION the update runs fastest on.  Why: the env_mlp backward level takes 39 us per
        optimizer step on some workspace allocations and 40 - 48 us on others of the same size and alignment -- constant for
        the life of the allocation (tools/probes/env_level_time.py), no other kernel of the step affected, and only in the
        context of a whole step (the launch repeated on its own, with its operands cache-resident, runs the fast figure on
        every allocation; identical TLB / L2 counters: DESIGN.md section 7) -- up to 2 % of the update.  So, once, with a
        rollout set: run ONE whole update on each of ``trials`` candidate workspaces (all alive at once, else the caching
        allocator hands the same block back), sum its kernels' durations from the library's dispatch timestamps, keep the
        fastest candidate, and put every state tensor and the step counter back exactly as they were (the workspace holds
        nothing that outlives an update).  No collectives: in a multi-rank job every rank tunes on its own.  Costs ``trials`` + 1
        updates of wall time; ``IGI_WS_TRIALS`` (default 6; 1 = off) sets the default.  Returns the per-candidate update
        durations in ms (first entry = the allocation the engine was built with), also kept in ``workspace_trial_ms``."""
        trials = int(os.environ.get("IGI_WS_TRIALS", "6")) if trials is None else int(trials)
        if trials <= 1 or self._ro is None or self.device.type != "cuda":
            return None
        keys = [k for k in ops.STATE_FIELDS if k not in ("perm", "workspace")]
        snap = {k: getattr(self, k).clone() for k in keys}
        t0, cfg0 = self.adam_t, (self.cfg.gamma, self.cfg.tau, self.cfg.lr)
        cands = [self.workspace]
        for _ in range(trials - 1):
            try:
                cands.append(torch.zeros_like(self.workspace))
            except RuntimeError:          # out of memory: choose among what there is
                break
        if len(cands) < 2:
            return None
        times = []
        try:
            with torch.cuda.device(self.device):
                self.prepare()
                self.update()            # untimed: the first update of a process also pays code loading and clock ramp-up
                for w in cands:
                    for k in keys:
                        getattr(self, k).copy_(snap[k])
                    self.adam_t, self.workspace = t0, w
                    self.prepare()
                    torch.cuda.synchronize()
                    _lib.prof_enable(True)
                    try:
                        self.update()
                        torch.cuda.synchronize()
                        classes = _lib.prof_read()
                    finally:
                        _lib.prof_enable(False)
                    times.append(round(sum(c["total_ms"] for c in classes), 4))
        finally:
            for k in keys:
                getattr(self, k).copy_(snap[k])
            self.adam_t = t0
            self.cfg.gamma, self.cfg.tau, self.cfg.lr = cfg0
            self.workspace = cands[0]
        if len(times) == len(cands) and all(t > 0 for t in times):
            self.workspace = cands[min(range(len(times)), key=times.__getitem__)]
            self.workspace_trial_ms = times
        return self.workspace_trial_ms

    def prepare(self, ro=None):
        """computer_return + prepare_training + value normalisation (experience.py:242-263;
        frozen_ppo.py:717-725)."""
        if ro is not None:
            self.set_rollout(ro)
        torch.ops.mi355ppo.gae_advnorm(self._ro, self.state_list(), *self._cfg_args(), bool(self.hp["normalize_value"]))

    def fwd_bwd(self, mb_index, slot):
        torch.ops.mi355ppo.ppo_minibatch_fwd_bwd(self._ro, self.state_list(), *self._cfg_args(), mb_index, slot, -1)

    def fwd_bwd_phase(self, mb_index, slot, phase):
        """Phase 0: down to dZ of the first trunk layer (the EARLY gradient bucket is final); phase 1: latent + env_mlp
        backward and the first trunk layer's weight gradient (the LATE bucket is final).  See ``grad_buckets``."""
        torch.ops.mi355ppo.ppo_minibatch_fwd_bwd(self._ro, self.state_list(), *self._cfg_args(), mb_index, slot, phase)

    @property
    def grad_buckets(self):
        """((early ranges), (late ranges)) of the flat gradient as (offset, length) pairs, empty ranges dropped:
        early = actor layers >= 1 | critic layers >= 1 + value + mu; late = sigma + env_mlp + actor layer 0 | critic
        layer 0 (igi_teacher_grad_buckets)."""
        off, ln = (C.c_int64 * 4)(), (C.c_int64 * 4)()
        n = self.L.igi_teacher_grad_buckets(C.byref(self.cfg), off, ln)
        if n != 4:
            _lib.check(n, "igi_teacher_grad_buckets")
        r = [(int(off[i]), int(ln[i])) for i in range(4)]
        return tuple(x for x in r[:2] if x[1] > 0), tuple(x for x in r[2:] if x[1] > 0)

    def bucket_views(self):
        early, late = self.grad_buckets
        return [self.grads[o:o + n] for o, n in early], [self.grads[o:o + n] for o, n in late]

    def apply(self, slot, grad_scale=1.0):
        self.adam_t += 1
        torch.ops.mi355ppo.ppo_clip_adam(self.state_list(), *self._cfg_args(), slot, self.adam_t, float(grad_scale))

    def update(self):
        """mini_epochs x n_minibatch optimizer steps enqueued back to back (frozen_ppo.py:508-640).
        Returns the (E*n_mb, 8) stats tensor (device; no host sync here)."""
        torch.ops.mi355ppo.ppo_update(self._ro, self.state_list(), *self._cfg_args(), self.adam_t)
        self.adam_t += self.E * self.n_mb
        return self.stats

    def update_dp(self, all_reduce, world_size, all_reduce_async=None):
        """Same loop with a gradient all-reduce between backward and the optimizer
        (frozen_ppo.py:586-603): SUM over ranks, the 1/world is folded into the Adam kernel.

        ``all_reduce_async(t) -> work`` (``dist.all_reduce(t, async_op=True)``) enables the overlapped
        schedule: the early bucket (trunk layers >= 1 and the heads, 81 % of the bytes, two ranges) is reduced on the
        collective's stream while the latent / env_mlp backward still runs on the compute stream; ``work.wait()`` only orders the
        streams, the host never blocks.  Either way the whole update is ONE native call
        (igi_teacher_update_dp); the library calls back between the stages of a step."""
        early, late = self.bucket_views()
        pending = []

        def reducer(bucket, step):
            if bucket == 2:
                for w in pending:
                    if w is not None:
                        w.wait()
                pending.clear()
            elif all_reduce_async is not None:
                for view in (early if bucket == 0 else late):
                    pending.append(all_reduce_async(view))
            elif bucket == 1:                 # serial schedule: everything after backward, like the reference
                all_reduce(self.grads)

        h = ops.register_reducer(reducer)
        try:
            torch.ops.mi355ppo.ppo_update_dp(self._ro, self.state_list(), *self._cfg_args(), self.adam_t,
                                             1.0 / world_size, h)
        finally:
            ops.unregister_reducer(h)
        self.adam_t += self.E * self.n_mb
        return self.stats

    def update_dp_native(self, comm, overlap=True, want_stats_sum=False):
        """The data-parallel update with the gradient exchange issued by the library over its own RCCL communicator
        (``utils.dist.NativeComm``): one native call, no callback -- igi_teacher_update_dp_rccl.  ``want_stats_sum``:
        also returns the per-step statistics summed over the ranks (one collective per update)."""
        stats_sum = torch.empty_like(self.stats) if want_stats_sum else None
        torch.ops.mi355ppo.ppo_update_dp_rccl(self._ro, self.state_list(), *self._cfg_args(), self.adam_t,
                                              int(comm.handle), bool(overlap), stats_sum)
        self.adam_t += self.E * self.n_mb
        return (self.stats, stats_sum) if want_stats_sum else self.stats

    def infer(self, obs, priv, want_latent=False, normalize=True):
        """model_act forward without sampling (models_split.py:120-164).  normalize=True: raw inputs,
        normalised with the current running stats (eval mode); False: inputs already processed.
        Returns (mu, value_normalised[, latent])."""
        obs = obs.to(self.device, torch.float32).contiguous()
        priv = priv.to(self.device, torch.float32).contiguous()
        mu, val, lat = torch.ops.mi355ppo.actor_critic_infer(self.state_list(), *self._cfg_args(), obs, priv,
                                                              bool(normalize), bool(want_latent))
        return (mu, val, lat) if want_latent else (mu, val)

    # ---- reference-shaped accessors ------------------------------------------------------------
    def env_major(self, x):
        """(T,N,...) -> (N*T,...) like experience.py:39-46 (a copy; off the hot path)."""
        s = x.shape
        return x.transpose(0, 1).reshape(s[0] * s[1], *s[2:])

    def rms_dict(self, packed):
        d = (packed.numel() - 1) // 2
        return dict(running_mean=packed[:d], running_var=packed[d:2 * d], count=packed[2 * d])
