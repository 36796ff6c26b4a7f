"""Teacher PPO trainer with the reference's interface (algo/ppo/frozen_ppo.py:113-877):
``PPO(env, output_dif, full_config)`` with ``train / train_epoch / play_steps / model_act / test /
save / restore_train / restore_test / set_eval / set_train / write_stats``, plus ``policy_kl`` and
``AdaptiveScheduler``.  ``train.py`` builds it with ``eval(cfg.train.algo)(envs, output_dif,
full_config=cfg)`` and calls ``agent.train()`` unchanged.

What is different underneath (MI355X-first):
  * one PPO update = igi_teacher_prepare + E*E x (igi_teacher_fwd_bwd, [RCCL all-reduce],
    igi_teacher_apply) enqueued on the current HIP stream with NO host synchronisation inside the
    loop: per-step losses / KL live in a device stats table read back once per update
    (the reference syncs every step at frozen_ppo.py:569 and flushes the allocator at :622);
  * multi-GPU: the flat gradient vector *is* the all-reduce buffer (no torch.cat / copy-back,
    frozen_ppo.py:586-603) and the 1/world factor is folded into the Adam kernel.
"""
import os
import time

import torch
import torch.distributed as dist

from ..models.models_split import ActorCriticSplit as ActorCritic
from ..models.running_mean_std import RunningMeanStd
from .experience import ExperienceBuffer
from ...teacher_native import TeacherEngine
from ...utils.misc import AverageScalarMeter, multi_gpu_aggregate_stats


class _NullWriter:
    """Stand-in when tensorboardX is not installed: keeps the last value of every scalar."""

    def __init__(self, *_a, **_k):
        self.scalars = {}

    def add_scalar(self, tag, value, step=None):
        self.scalars[tag] = (value, step)


def log_test_result(log_file, **fields):
    """Append one evaluation record to ``stage*_nn/log.json`` (frozen_ppo.py:45-110 / ext_adapt.py:40-90 keep such a
    list with best / current reward, loss, agent steps, success rate and a timestamp; the plot is omitted)."""
    import json
    from datetime import datetime
    rec = {k: (v.item() if torch.is_tensor(v) else v) for k, v in fields.items()}
    rec['timestamp'] = datetime.now().isoformat()
    data = []
    if os.path.exists(log_file):
        try:
            with open(log_file) as f:
                data = json.load(f) or []
        except (ValueError, OSError):
            data = []
    data.append(rec)
    with open(log_file, 'w') as f:
        json.dump(data, f, indent=4)


def _summary_writer(path):
    try:
        from tensorboardX import SummaryWriter
        return SummaryWriter(path)
    except Exception:
        return _NullWriter()


class _FusedAdam:
    """Holds what torch.optim.Adam held for the reference (frozen_ppo.py:192-194); the update itself is
    the k_clip_adam kernel.  ``param_groups[0]['lr']`` is honoured (frozen_ppo.py:636-640)."""

    def __init__(self, engine, lr, weight_decay=0.0):
        if weight_decay:
            raise NotImplementedError("weight_decay is 0 in every reference config")
        self.engine = engine
        self.param_groups = [{"lr": float(lr)}]

    def state_dict(self):
        return {"exp_avg": self.engine.adam_m, "exp_avg_sq": self.engine.adam_v, "step": self.engine.adam_t,
                "lr": self.param_groups[0]["lr"]}

    def load_state_dict(self, sd):
        self.engine.adam_m.copy_(sd["exp_avg"])
        self.engine.adam_v.copy_(sd["exp_avg_sq"])
        self.engine.adam_t = int(sd["step"])
        self.param_groups[0]["lr"] = float(sd["lr"])


class PPO(object):
    def __init__(self, env, output_dif, full_config):
        # ---- MultiGPU (frozen_ppo.py:116-126)
        self.multi_gpu = full_config.train.ppo.multi_gpu
        if self.multi_gpu:
            from ...utils.dist import init_rank_device
            self.rank, self.rank_size, self.device = init_rank_device()   # "nccl" IS RCCL on ROCm
        else:
            self.rank = -1
            self.rank_size = 1
            self.device = full_config["rl_device"]
        self.full_config = full_config
        self.task_config = full_config.task
        self.network_config = full_config.train.network
        self.ppo_config = full_config.train.ppo
        self.env = env
        self.num_actors = self.ppo_config['num_actors']
        self.actions_num = self.task_config.env.numActions
        self.obs_shape = (self.task_config.env.numObservations * self.task_config.env.numObsHist,)
        self.vt_policy = False
        self.priv_info = self.ppo_config['priv_info']
        self.priv_info_dim = self.ppo_config['priv_info_dim']
        self.gt_contacts_info = self.ppo_config['compute_contact_gt']
        self.only_contact = self.ppo_config['only_contact']
        self.num_contacts_points = self.ppo_config['num_points']
        self.priv_info_embed_dim = self.network_config.priv_mlp.units[-1]
        net_config = {
            'actor_units': self.network_config.mlp.units, 'actions_num': self.actions_num,
            'input_shape': self.obs_shape, 'priv_mlp_units': self.network_config.priv_mlp.units,
            'priv_info_dim': self.priv_info_dim, 'priv_info': self.priv_info,
            'gt_contacts_info': self.gt_contacts_info, 'only_contact': self.only_contact,
            'contacts_mlp_units': self.network_config.contact_mlp.units,
            'num_contact_points': self.num_contacts_points,
            'shared_parameters': self.ppo_config.shared_parameters, 'full_config': self.full_config,
            'vt_policy': self.vt_policy,
        }
        self.model = ActorCritic(net_config)
        self.model.to(self.device)
        self.running_mean_std = RunningMeanStd(self.obs_shape).to(self.device)
        self.priv_mean_std = RunningMeanStd((self.priv_info_dim,)).to(self.device)
        self.value_mean_std = RunningMeanStd((1,)).to(self.device)

        self.output_dir = output_dif
        self.extra_info = {}
        self.writer = _NullWriter()
        if env is not None and not full_config.get('offline_training', False) and output_dif is not None:
            self.nn_dir = os.path.join(self.output_dir, 'stage1_nn')
            self.tb_dif = os.path.join(self.output_dir, 'stage1_tb')
            os.makedirs(self.nn_dir, exist_ok=True)
            os.makedirs(self.tb_dif, exist_ok=True)
            self.writer = _summary_writer(self.tb_dif)

        # ---- PPO hyper-parameters (frozen_ppo.py:191-220)
        self.last_lr = float(self.ppo_config['learning_rate'])
        self.weight_decay = self.ppo_config.get('weight_decay', 0.0)
        self.e_clip = self.ppo_config['e_clip']
        self.clip_value = self.ppo_config['clip_value']
        self.entropy_coef = self.ppo_config['entropy_coef']
        self.critic_coef = self.ppo_config['critic_coef']
        self.bounds_loss_coef = self.ppo_config['bounds_loss_coef']
        self.gamma = self.ppo_config['gamma']
        self.tau = self.ppo_config['tau']
        self.truncate_grads = self.ppo_config['truncate_grads']
        self.grad_norm = self.ppo_config['grad_norm']
        self.value_bootstrap = self.ppo_config['value_bootstrap']
        self.normalize_advantage = self.ppo_config['normalize_advantage']
        self.normalize_input = self.ppo_config['normalize_input']
        self.normalize_value = self.ppo_config['normalize_value']
        if not (self.normalize_input and self.normalize_advantage and self.clip_value):
            raise NotImplementedError("the fused update implements the reference defaults: "
                                      "normalize_input / normalize_advantage / clip_value = True")
        self.horizon_length = self.ppo_config['horizon_length']
        self.batch_size = self.horizon_length * self.num_actors
        self.mini_epochs_num = self.ppo_config['mini_epochs']
        self.minibatch_size = self.batch_size // self.mini_epochs_num  # YAML minibatch_size ignored (:215)
        assert self.batch_size % self.minibatch_size == 0 or full_config.test
        self.kl_threshold = self.ppo_config['kl_threshold']
        self.scheduler = AdaptiveScheduler(self.kl_threshold)
        self.save_freq = self.ppo_config['save_frequency']
        self.save_best_after = self.ppo_config['save_best_after']
        self.it = 0
        self.episode_rewards = AverageScalarMeter(100)
        self.episode_lengths = AverageScalarMeter(100)
        self.episode_success = AverageScalarMeter(100)
        self.obs = None
        self.epoch_num = 0

        self.storage = ExperienceBuffer(self.num_actors, self.horizon_length, self.batch_size,
                                        self.minibatch_size, self.obs_shape[0], self.actions_num,
                                        self.priv_info_dim, self.num_contacts_points, self.vt_policy, self.device)

        # ---- native engine: owns the flat parameter / gradient / Adam vectors and the workspace;
        #      model parameters and normaliser buffers become views into it
        self.engine = TeacherEngine(
            self.num_actors, self.horizon_length, self.mini_epochs_num, units=list(self.network_config.mlp.units),
            priv_units=list(self.network_config.priv_mlp.units), obs_dim=self.obs_shape[0],
            priv_dim=self.priv_info_dim, act_dim=self.actions_num, device=self.device, perm=self.storage.indices,
            gamma=self.gamma, tau=self.tau, lr=self.last_lr, e_clip=self.e_clip, critic_coef=self.critic_coef,
            entropy_coef=self.entropy_coef, bounds_loss_coef=self.bounds_loss_coef, grad_norm=self.grad_norm,
            truncate_grads=self.truncate_grads, normalize_value=self.normalize_value)
        self.model.bind_flat_to(self.engine)
        self.model.attach_engine(self.engine)
        self.running_mean_std.bind(self.engine.rms_obs)
        self.priv_mean_std.bind(self.engine.rms_priv)
        self.value_mean_std.bind(self.engine.rms_value)
        self.storage.attach(self.engine)
        self.optimizer = _FusedAdam(self.engine, self.last_lr, self.weight_decay)

        batch_size = self.num_actors
        self.current_rewards = torch.zeros((batch_size, 1), dtype=torch.float32, device=self.device)
        self.current_lengths = torch.zeros(batch_size, dtype=torch.float32, device=self.device)
        self.current_success = torch.zeros(batch_size, dtype=torch.float32, device=self.device)
        self.dones = torch.ones((batch_size,), dtype=torch.uint8, device=self.device)
        self.agent_steps = 0
        self.max_agent_steps = self.ppo_config['max_agent_steps']
        self.best_rewards = -10000
        self.cur_reward = self.best_rewards
        self.success_rate = 0
        self.data_collect_time = 0
        self.rl_train_time = 0
        self.all_time = 0

    # ------------------------------------------------------------------------------------------
    def write_stats(self, a_losses, c_losses, b_losses, entropies, kls, grad_norms, returns_list):
        """frozen_ppo.py:279-318"""
        w = self.writer
        w.add_scalar('performance/RLTrainFPS', self.agent_steps / max(self.rl_train_time, 1e-9), self.agent_steps)
        w.add_scalar('performance/EnvStepFPS', self.agent_steps / max(self.data_collect_time, 1e-9), self.agent_steps)
        mean = lambda xs: (torch.mean(torch.stack(xs)) if isinstance(xs, list) else torch.mean(xs)).item()
        w.add_scalar('losses/actor_loss', mean(a_losses), self.agent_steps)
        w.add_scalar('losses/bounds_loss', mean(b_losses), self.agent_steps)
        w.add_scalar('losses/critic_loss', mean(c_losses), self.agent_steps)
        w.add_scalar('losses/entropy', mean(entropies), self.agent_steps)
        w.add_scalar('info/kl', mean(kls), self.agent_steps)
        w.add_scalar('info/grad_norms', mean(grad_norms), self.agent_steps)
        w.add_scalar('info/last_lr', self.last_lr, self.agent_steps)
        w.add_scalar('info/e_clip', self.e_clip, self.agent_steps)
        if returns_list:
            w.add_scalar('info/returns_list', mean(returns_list), self.agent_steps)
        for k, v in self.extra_info.items():
            w.add_scalar(f'{k}', v, self.agent_steps)

    def set_eval(self):
        self.model.eval()
        self.running_mean_std.eval()
        self.priv_mean_std.eval()
        self.value_mean_std.eval()

    def set_train(self):
        self.model.train()
        self.running_mean_std.train()
        self.priv_mean_std.train()
        self.value_mean_std.train()

    @torch.no_grad()
    def model_act(self, obs_dict):
        """frozen_ppo.py:343-366: input normalisation (eval statistics), network forward, Gaussian
        sample, value de-normalisation -- one native forward (igi_teacher_infer) + sampling."""
        eng = self.engine
        f32 = dict(dtype=torch.float32, device=self.device)
        obs = obs_dict['obs'].to(**f32).contiguous()
        priv = obs_dict['priv_info'].to(**f32).contiguous()
        n, a = obs.shape[0], self.actions_num
        actions, mu, sigma, clamped = (torch.empty((n, a), **f32) for _ in range(4))
        nlp, values, values_out = torch.empty(n, **f32), torch.empty((n, 1), **f32), torch.empty((n, 1), **f32)
        noise = torch.randn_like(mu)
        # ONE native call: normalise, env_mlp, trunk, heads, sample, neglogp, value de-normalisation
        torch.ops.mi355ppo.rollout_policy_step(eng.state_list(), *eng._cfg_args(), obs, priv, True, noise,
                                               self.value_mean_std._packed if self.normalize_value else None,
                                               None, None, actions, nlp, values, mu, sigma, clamped, values_out)
        return {'neglogpacs': nlp, 'values': values, 'actions': actions, 'mus': mu, 'sigmas': sigma}

    # ------------------------------------------------------------------------------------------
    def train(self):
        """frozen_ppo.py:368-446"""
        _t = time.time()
        _last_t = time.time()
        self.obs = self.env.reset(reset_at_success=False, reset_at_fails=True)
        test_every = getattr(self, 'test_every', 10e6)      # frozen_ppo.py:372-373
        self.next_test_step = test_every
        self.agent_steps = self.batch_size if not self.multi_gpu else self.batch_size * self.rank_size
        if self.multi_gpu:
            dist.broadcast(self.engine.params, 0)   # tensor broadcast instead of pickled state_dict (:376-381)
        while self.agent_steps < self.max_agent_steps:
            self.epoch_num += 1
            a_losses, c_losses, b_losses, entropies, kls, grad_norms, returns_list = self.train_epoch()
            self.storage.data_dict = None
            if self.multi_gpu:
                a_losses, b_losses, c_losses, entropies, kls, grad_norms = multi_gpu_aggregate_stats(
                    [a_losses, b_losses, c_losses, entropies, kls, grad_norms])
                mean_rewards, mean_lengths, mean_success = multi_gpu_aggregate_stats(
                    [torch.tensor([m.get_mean()], dtype=torch.float32, device=self.device)
                     for m in (self.episode_rewards, self.episode_lengths, self.episode_success)])
            else:
                mean_rewards = self.episode_rewards.get_mean()
                mean_lengths = self.episode_lengths.get_mean()
                mean_success = self.episode_success.get_mean()
            if not self.multi_gpu or self.rank == 0:
                all_fps = self.agent_steps / (time.time() - _t)
                last_fps = self.batch_size / (time.time() - _last_t)
                _last_t = time.time()
                print(f'Agent Steps: {int(self.agent_steps // 1e6):04}M | FPS: {all_fps:.1f} | '
                      f'Last FPS: {last_fps:.1f} | Collect Time: {self.data_collect_time / 60:.1f} min | '
                      f'Train RL Time: {self.rl_train_time / 60:.1f} min | Best Reward: {self.best_rewards:.2f} | '
                      f'Cur Reward: {mean_rewards:.2f}')
                self.cur_reward = mean_rewards
                self.write_stats(a_losses, c_losses, b_losses, entropies, kls, grad_norms, returns_list)
                self.writer.add_scalar('episode_rewards/step', mean_rewards, self.agent_steps)
                self.writer.add_scalar('episode_lengths/step', mean_lengths, self.agent_steps)
                self.writer.add_scalar('mean_success/step', mean_success, self.agent_steps)
                if self.agent_steps >= self.next_test_step:     # frozen_ppo.py:422-430: evaluate, save 'last'
                    self.test(total_steps=self.env.cfg_task.rl.max_episode_length)
                    self.obs = self.env.reset(reset_at_success=False, reset_at_fails=True)
                    self.set_train()
                    self.next_test_step += test_every
                    if self.output_dir is not None:
                        self.save(os.path.join(self.nn_dir, 'last'))
                if mean_rewards > self.best_rewards and self.agent_steps >= self.save_best_after \
                        and mean_rewards != 0.0 and self.output_dir is not None:
                    prev = os.path.join(self.nn_dir, f"best_reward_{self.best_rewards:.2f}.pth")
                    if os.path.exists(prev):
                        os.remove(prev)
                    self.best_rewards = mean_rewards
                    self.save(os.path.join(self.nn_dir, f"best_reward_{mean_rewards:.2f}"))
                self.success_rate = mean_success
        print('max steps achieved')

    def save(self, name):
        """frozen_ppo.py:448-463: same keys / dtypes, so .pth files interchange with the reference."""
        weights = {'model': self.model.state_dict(),
                   'running_mean_std': self.running_mean_std.state_dict(),
                   'priv_mean_std': self.priv_mean_std.state_dict(),
                   'value_mean_std': self.value_mean_std.state_dict()}
        torch.save(weights, f'{name}.pth')

    def restore_train(self, fn, *args, **kwargs):
        """frozen_ppo.py:465-475; also tolerates train.py:136's positional call.  Like the
        reference, the checkpoint's value_mean_std is NOT restored (a resumed run re-learns the value statistics)."""
        if not fn:
            return
        checkpoint = torch.load(fn, map_location=self.device)
        self.model.load_state_dict(checkpoint['model'])
        self.running_mean_std.load_state_dict(checkpoint['running_mean_std'])
        self.priv_mean_std.load_state_dict(checkpoint['priv_mean_std'])

    def restore_test(self, fn):
        """frozen_ppo.py:477-484"""
        checkpoint = torch.load(fn, map_location=self.device)
        self.model.load_state_dict(checkpoint['model'])
        if self.normalize_input:
            self.running_mean_std.load_state_dict(checkpoint['running_mean_std'])
            self.priv_mean_std.load_state_dict(checkpoint['priv_mean_std'])

    def _native_comm(self):
        """The library's RCCL communicator of this rank (utils.dist.NativeComm), created on first use when the process
        group runs over RCCL; None under any other backend (gloo in the CPU-side / single-GPU tests) or with
        IGI_DP_NATIVE=0."""
        if not hasattr(self, "_comm"):
            self._comm = None
            from ...utils.dist import native_comm_or_none
            self._comm = native_comm_or_none(self.device, self.rank_size)
        return self._comm

    # ------------------------------------------------------------------------------------------
    def update(self):
        """The optimisation half of train_epoch (frozen_ppo.py:503-646) on the rollout currently in
        storage (prepare_training already run).  Returns the reference's seven lists."""
        eng = self.engine
        eng.cfg.lr = float(self.optimizer.param_groups[0]["lr"])
        stats_sum = None
        if self.multi_gpu and self._native_comm() is not None:
            # the library issues both bucket all-reduces itself (own RCCL communicator + communication stream) and
            # returns the statistics summed over the ranks: no Python between the 64 steps, no separate KL collective
            stats, stats_sum = eng.update_dp_native(self._comm, overlap=os.environ.get("IGI_DP_OVERLAP", "1") != "0",
                                                     want_stats_sum=True)
        elif self.multi_gpu:
            # torch.distributed callback path (gloo, tests): two gradient buckets; the large one is reduced while the
            # env_mlp backward still runs
            stats = eng.update_dp(lambda t: dist.all_reduce(t, op=dist.ReduceOp.SUM), self.rank_size,
                                  all_reduce_async=lambda t: dist.all_reduce(t, op=dist.ReduceOp.SUM, async_op=True))
        else:
            stats = eng.update()
        E, n_mb = self.mini_epochs_num, len(self.storage)
        s = stats[:E * n_mb]
        a_losses, c_losses, b_losses = list(s[:, 0].unbind()), list(s[:, 1].unbind()), list(s[:, 2].unbind())
        entropies, grad_norms = list(s[:, 3].unbind()), list(s[:, 6].unbind())
        if stats_sum is not None:               # frozen_ppo.py:624-627: mean KL over the ranks, from the summed rows
            av_kls = stats_sum[:E * n_mb, 4].reshape(E, n_mb).mean(dim=1) / self.rank_size
        else:
            av_kls = s[:, 4].reshape(E, n_mb).mean(dim=1)
            if self.multi_gpu:                  # one collective instead of E
                dist.all_reduce(av_kls, op=dist.ReduceOp.SUM)
                av_kls = av_kls / self.rank_size
        kls = list(av_kls.unbind())
        for pg in self.optimizer.param_groups:  # lr is constant: scheduler.update is commented out (:630)
            pg["lr"] = self.last_lr
        returns_list = [self.engine.returns_n.mean()]
        return a_losses, c_losses, b_losses, entropies, kls, grad_norms, returns_list

    def train_epoch(self):
        """frozen_ppo.py:495-646"""
        _t = time.time()
        self.set_eval()
        self.play_steps()
        torch.cuda.synchronize(self.device)
        self.data_collect_time += (time.time() - _t)
        _t = time.time()
        self.set_train()
        out = self.update()
        torch.cuda.synchronize(self.device)
        self.rl_train_time += (time.time() - _t)
        return out

    def play_steps(self):
        """frozen_ppo.py:648-725.  Per environment step: the generator's noise, ONE native policy step
        (igi_rollout_policy_step: normalise + forward + sample / neglogp / value de-normalisation / arena writes, 7
        launches) and one bookkeeping launch after env.step (igi_rollout_env_store: dones, shaped reward, episode
        accumulators and the meters' sums) -- no host read-back inside the loop (the reference gathers finished episodes with
        ``nonzero`` every step, :693-696).  Pinned to the reference's own play_steps by tests/test_gpu_rollout.py."""
        sd = self.storage.storage_dict
        N, A = self.num_actors, self.actions_num
        T = self.horizon_length
        f32 = dict(dtype=torch.float32, device=self.device)
        meter = torch.zeros((T, 4), **f32)
        clamped = torch.empty((N, A), **f32)
        values = torch.empty((N, 1), **f32)
        rms_v = self.value_mean_std._packed if self.normalize_value else None   # frozen_ppo.py:364-365
        policy_step, env_store = torch.ops.mi355ppo.rollout_policy_step, torch.ops.mi355ppo.rollout_env_store
        state, (icfg, fcfg) = self.engine.state_list(), self.engine._cfg_args()
        for n in range(T):
            self.it += 1
            obs = self.obs['obs'].to(**f32).contiguous()
            priv = self.obs['priv_info'].to(**f32).contiguous()
            noise = torch.randn_like(clamped)       # the reference's draw: Normal sampling is mu + sigma * randn_like(mu)
            # policy forward + sampling + arena writes of this step: one native call (igi_rollout_policy_step)
            policy_step(state, icfg, fcfg, obs, priv, True, noise, rms_v, sd['obses'][n], sd['priv_info'][n],
                        sd['actions'][n], sd['neglogpacs'][n], sd['values'][n], sd['mus'][n], sd['sigmas'][n],
                        clamped, values)
            self.obs, rewards, self.dones, infos = self.env.step(clamped)
            assert isinstance(infos, dict), 'Info Should be a Dict'
            rewards = rewards.to(**f32).contiguous()
            dones = self.dones if self.dones.dtype == torch.uint8 else self.dones.to(torch.uint8)
            touts = infos.get('time_outs') if self.value_bootstrap else None
            if touts is not None:
                touts = (touts.view(torch.uint8) if touts.dtype == torch.bool else touts.to(torch.uint8)).contiguous()
            succ = infos.get('successes')
            succ = succ.to(**f32).contiguous() if succ is not None else None
            env_store(rewards, dones.contiguous(), values, touts, succ, float(self.gamma), touts is not None,
                      sd['rewards'][n], sd['dones'][n], self.current_rewards, self.current_lengths,
                      self.current_success, meter[n])
            self.extra_info = {k: v for k, v in infos.items()
                               if isinstance(v, (float, int)) or (isinstance(v, torch.Tensor) and v.dim() == 0)}
        self.episode_rewards.update_sums(meter[:, 0], meter[:, 3])
        self.episode_lengths.update_sums(meter[:, 1], meter[:, 3])
        self.episode_success.update_sums(meter[:, 2], meter[:, 3])
        self._finish_rollout()

    def _finish_rollout(self):
        last_values = self.model_act(self.obs)['values']
        self.agent_steps = (self.agent_steps + self.batch_size) if not self.multi_gpu \
            else self.agent_steps + self.batch_size * self.rank_size
        self.storage.computer_return(last_values, self.gamma, self.tau)
        # frozen_ppo.py:715-725: prepare_training + the two value_mean_std updates, fused
        self.storage.prepare_training(self.value_mean_std if self.normalize_value else None)

    @torch.no_grad()
    def test(self, milestone=100, total_steps=1e9):
        """frozen_ppo.py:727-789: roll the deterministic policy (eval-mode normalisers, clamped mean action) for
        one episode clock without resets at success and return (num_success, total_dones) from the env's
        ``success_reset_buf``; a new best success rate is checkpointed as best_succ_*.  Video / plot / trajectory
        logging side effects belong to the simulator side and are omitted."""
        self.set_eval()
        obs = self.env.reset(reset_at_success=False, reset_at_fails=False)
        steps, last = 0, int(getattr(self.env, 'max_episode_length', total_steps)) - 1
        while steps < min(total_steps, last):
            steps += 1
            mu, _ = self.engine.infer(obs['obs'], obs['priv_info'], normalize=True)
            obs, r, done, info = self.env.step(torch.clamp(mu, -1.0, 1.0))
        if not hasattr(self.env, 'test_reset_buf'):
            return 0, 0
        finished = self.env.test_reset_buf > 0
        num_success = int((self.env.success_reset_buf * finished).sum().item())
        total_dones = int(finished.sum().item())
        self.test_success = num_success / max(total_dones, 1)
        if self.output_dir is not None:
            log_test_result(os.path.join(self.nn_dir, 'log.json'), best_reward=self.best_rewards,
                            cur_reward=getattr(self, 'cur_reward', 0.0), steps=self.agent_steps,
                            success_rate=self.test_success)
        if self.output_dir is not None and self.test_success > getattr(self, 'best_success', -1.0) \
                and self.agent_steps > 1e5:
            self.best_success = self.test_success
            self.save(os.path.join(self.nn_dir, f'best_succ_{self.best_success:.2f}'))
        return num_success, total_dones


def policy_kl(p0_mu, p0_sigma, p1_mu, p1_sigma):
    """frozen_ppo.py:854-860"""
    c1 = torch.log(p1_sigma / p0_sigma + 1e-5)
    c2 = (p0_sigma ** 2 + (p1_mu - p0_mu) ** 2) / (2.0 * (p1_sigma ** 2 + 1e-5))
    kl = (c1 + c2 - 0.5).sum(dim=-1)
    return kl.mean()


class AdaptiveScheduler(object):
    """frozen_ppo.py:864-877 (constructed but never stepped: SURVEY Appendix A5)."""

    def __init__(self, kl_threshold=0.008):
        self.min_lr = 1e-6
        self.max_lr = 1e-2
        self.kl_threshold = kl_threshold

    def update(self, current_lr, kl_dist):
        lr = current_lr
        if kl_dist > (2.0 * self.kl_threshold):
            lr = max(current_lr / 1.5, self.min_lr)
        if kl_dist < (0.5 * self.kl_threshold):
            lr = min(current_lr * 1.5, self.max_lr)
        return lr
