"""Online student distillation with the reference's interface (algo/ext_adapt/ext_adapt.py:169-1232):
``ExtrinsicAdapt(env, output_dir, full_config)`` with ``process_obs / play_steps / train_epoch / train /
save / restore_train / restore_test / set_eval / set_student_eval / set_student_train``.

The frozen teacher acts through igi_teacher_infer, the student's tactile CNN and PointNets run as HIP
autograd ops, normalisers are igi_rms_forward, the optimizer is one native clip+Adam pass over a flat
buffer whose gradient half is also the RCCL all-reduce buffer (SUM, 1/world folded into the step) --
replacing the reference's cat / all_reduce / copy-back (ext_adapt.py:833-851).  ``only_bc=True`` (every
stage-2/3 launch script of the reference): the student emits the action.  ``only_bc=False``: the student emits
the 8-d latent and the frozen teacher's actor maps ``cat(obs, latent)`` to the action with autograd through the
native Linear+Tanh op (``ActorCriticSplit.act_with_grad``), ext_adapt.py:799-806.
"""
import os
import time

import torch
import torch.distributed as dist

from ..models.models_split import ActorCriticSplit as ActorCritic
from ..models.running_mean_std import RunningMeanStd
from ..models.transformer.runner import Runner as Student
from ..ppo.experience import StudentBuffer
from ..ppo.frozen_ppo import _NullWriter, _summary_writer, log_test_result
from ...bc_loss import bc_loss
from ...optim import FlatAdam
from ...utils.misc import AverageScalarMeter


class ExtrinsicAdapt(object):
    def __init__(self, env, output_dir, full_config):
        self.multi_gpu = full_config.train.ppo.multi_gpu
        if self.multi_gpu:
            from ...utils.dist import init_rank_device
            self.rank, self.rank_size, self.device = init_rank_device()
        else:
            self.rank = -1
            self.rank_size = 1
            self.device = full_config["rl_device"]
        self.full_config = full_config
        self.task_config = full_config.task
        self.network_config = full_config.train.network
        self.train_config = full_config.offline_train
        self.ppo_config = full_config.train.ppo
        self.only_bc = self.train_config.only_bc
        self.env = env
        self.num_actors = self.ppo_config['num_actors']
        self.obs_shape = (self.task_config.env.numObservations * self.task_config.env.numObsHist,)
        self.obs_stud_shape = (self.task_config.env.numObsStudent * self.task_config.env.numObsStudentHist,)
        self.max_agent_steps = self.ppo_config['max_agent_steps']
        self.actions_num = self.task_config.env.numActions
        self.obs_info = self.ppo_config["obs_info"]
        self.tactile_info = self.ppo_config["tactile_info"]
        self.img_info = self.ppo_config["img_info"]
        self.seg_info = self.ppo_config["seg_info"]
        self.pcl_info = self.ppo_config["pcl_info"]
        self.priv_info = self.ppo_config['priv_info']
        self.priv_info_dim = self.ppo_config['priv_info_dim']
        agent_config = {
            'actor_units': self.network_config.mlp.units, 'actions_num': self.actions_num,
            'priv_mlp_units': self.network_config.priv_mlp.units, 'input_shape': self.obs_shape,
            'priv_info_dim': self.priv_info_dim, 'priv_info': self.priv_info,
            'gt_contacts_info': self.ppo_config['compute_contact_gt'],
            'only_contact': self.ppo_config['only_contact'],
            'contacts_mlp_units': self.network_config.contact_mlp.units,
            'num_contact_points': self.ppo_config['num_points'],
            'shared_parameters': self.ppo_config.shared_parameters, 'full_config': full_config, 'vt_policy': False,
        }
        self.agent = ActorCritic(agent_config)
        self.agent.to(self.device)
        self.agent.eval()
        self.running_mean_std = RunningMeanStd(self.obs_shape).to(self.device)
        self.running_mean_std.eval()
        self.priv_mean_std = RunningMeanStd((self.priv_info_dim,)).to(self.device)
        self.priv_mean_std.eval()
        student_cfg = self.full_config
        student_cfg.offline_train.model.use_tactile = self.tactile_info
        student_cfg.offline_train.model.use_seg = self.seg_info
        student_cfg.offline_train.model.use_lin = self.obs_info
        student_cfg.offline_train.model.use_img = self.img_info
        student_cfg.offline_train.model.use_pcl = self.pcl_info
        self.stud_obs_mean_std = RunningMeanStd(self.obs_stud_shape).to(self.device)
        self.stud_obs_mean_std.train()
        self.pcl_mean_std = RunningMeanStd((3,)).to(self.device)
        self.pcl_mean_std.train()
        self.student = Student(student_cfg)
        self.stats = None
        self.output_dir = output_dir
        self.writer = _NullWriter()
        if output_dir is not None:
            self.nn_dir = os.path.join(self.output_dir, 'stage2_nn')
            self.tb_dir = os.path.join(self.output_dir, 'stage2_tb')
            os.makedirs(self.nn_dir, exist_ok=True)
            os.makedirs(self.tb_dir, exist_ok=True)
            self.writer = _summary_writer(self.tb_dir)
        self.direct_info = {}
        self.horizon_length = self.ppo_config['horizon_length']
        self.batch_size = self.horizon_length * self.num_actors
        self.mini_epochs_num = self.ppo_config['mini_epochs']
        self.minibatch_size = self.batch_size // self.mini_epochs_num
        assert self.batch_size % self.minibatch_size == 0
        student_shapes = {'img': self.env.img_queue.shape[1:] if self.img_info else None,
                          'seg': self.env.seg_queue.shape[1:] if self.seg_info else None,
                          'tactile': self.env.tactile_queue.shape[1:] if self.tactile_info else None,
                          'student_obs': self.obs_stud_shape[0] if self.obs_info else None,
                          'pcl': self.env.pcl_queue.shape[1:] if self.pcl_info else None}
        self.storage = StudentBuffer(self.num_actors, self.horizon_length, self.batch_size, self.minibatch_size,
                                     self.obs_shape[0], self.actions_num, self.priv_info_dim, student_shapes,
                                     self.device)
        self.mean_eps_reward = AverageScalarMeter(window_size=100)
        self.mean_eps_length = AverageScalarMeter(window_size=100)
        self.mean_eps_success = AverageScalarMeter(window_size=100)
        self.latent_scale = self.train_config.train.latent_scale
        self.action_scale = self.train_config.train.action_scale
        self.best_rewards = -10000
        self.best_loss = 10000
        self.cur_reward = self.best_rewards
        self.agent_steps = 0
        for _, p in self.agent.named_parameters():        # teacher frozen (ext_adapt.py:304-305)
            p.requires_grad = False
        # Adam(lr 3e-4) + clip 0.5 (ext_adapt.py:307, 853) on a flat buffer
        # the encoders sit at the bottom of the graph: their gradients are final last, the decoder side's first -- the
        # flat gradient is laid out [encoders | decoder side] so that the latter can go to the gradient exchange while
        # the encoders' backward still runs (update())
        late = [p for n, p in self.student.model.named_parameters()
                if n.split('.')[0] in ('tactile_encoder', 'img_encoder', 'seg_encoder', 'pcl_encoder', 'lin_encoder')]
        self.optim = FlatAdam(self.student.model.parameters(), lr=3e-4, max_norm=0.5, late=late)
        batch_size = self.num_actors
        self.step_reward = torch.zeros((batch_size, 1), dtype=torch.float32, device=self.device)
        self.step_length = torch.zeros(batch_size, dtype=torch.float32, device=self.device)
        self.step_success = torch.zeros(batch_size, dtype=torch.float32, device=self.device)
        self.it = 0
        self.loss_weights = torch.ones(6, device=self.device)
        self.loss_weights[2] = 0.1                        # ext_adapt.py:812-813
        self.grad_probe = None                            # optional callable(step, model) between backward and the optimizer
        self.obs = None

    # ------------------------------------------------------------------------------------------
    def set_eval(self):
        self.agent.eval()
        self.running_mean_std.eval()
        self.priv_mean_std.eval()

    def set_student_eval(self):
        self.student.model.eval()

    def set_student_train(self):
        """ext_adapt.py:366-375"""
        self.student.model.train()
        if not self.train_config.from_offline:
            self.stud_obs_mean_std.train()
            self.pcl_mean_std.train()

    def process_obs(self, obs, obj_id=2, socket_id=3, distinct=True):
        """ext_adapt.py:383-435: normalise student_obs and the point cloud (both normalisers are in
        train mode during rollout: SURVEY Appendix A17); tactile passes through."""
        student_obs = obs['student_obs'] if self.obs_info else None
        tactile = obs['tactile'] if self.tactile_info else None
        img = obs['img'] if self.img_info else None
        seg = obs['seg'] if self.seg_info else None
        pcl = obs['pcl'] if self.pcl_info else None
        if self.seg_info:                       # keep plug + socket pixels only (ext_adapt.py:391-396)
            valid_mask = ((seg == obj_id) | (seg == socket_id)).float()
            seg = seg * valid_mask if distinct else valid_mask
            if self.img_info:
                img = img * valid_mask
        if self.pcl_info:
            pcl = self.pcl_mean_std(pcl.reshape(-1, 3)).reshape((obs['pcl'].shape[0], -1, 3))
        if student_obs is not None:
            if self.stats is not None and self.train_config.from_offline:
                # a student pre-trained offline keeps the dataset's statistics (ext_adapt.py:411-417):
                # [eef position + 6-D rotation (9) | socket position (3) | previous action]
                m, sd = self.stats["mean"], self.stats["std"]
                eef = (student_obs[:, :9] - m['eef_pos_rot6d']) / sd['eef_pos_rot6d']
                socket = (student_obs[:, 9:12] - m["socket_pos"][:3]) / sd["socket_pos"][:3]
                student_obs = torch.cat([eef, socket, student_obs[:, 12:]], dim=-1)
            elif not self.train_config.from_offline:
                student_obs = self.stud_obs_mean_std(student_obs)
            else:
                raise RuntimeError("from_offline=True needs restore_student(..., from_offline=True) first "
                                   "(it loads normalization.pkl)")
        return {'student_obs': student_obs, 'tactile': tactile, 'img': img, 'seg': seg, 'pcl': pcl}

    def play_latent_step(self, obs_dict):
        """The hook ``Runner(cfg, agent, action_regularization=True)`` asks its agent for (runner.py:37, 237-240,
        336-339): the frozen teacher's actor applied to ``cat(normalised obs, student latent)`` with autograd back
        into the latent, so an offline student trained on latents can also be penalised on the resulting action.
        (No agent class of the reference defines it; this is the behaviour its call sites assume.)"""
        return self.agent.act_with_grad({'obs': self.running_mean_std(obs_dict['obs']), 'latent': obs_dict['latent']})

    @torch.no_grad()
    def student_act(self, obs_dict):
        """One closed-loop student step (ext_adapt.py:585-607): process_obs -> student -> (frozen actor when the
        student emits a latent) -> clamp.  Returns (action in [-1, 1], latent)."""
        latent, _ = self.student.predict(self.process_obs(obs_dict), requires_grad=False)
        if not self.only_bc:
            mu, latent = self.agent.act_inference({'obs': self.running_mean_std(obs_dict['obs']), 'latent': latent})
        else:
            mu = latent
        return torch.clamp(mu, -1.0, 1.0), latent

    @torch.no_grad()
    def test(self, total_steps=1e9):
        """ext_adapt.py:563-656: roll the STUDENT in the environment without resets at success until every
        episode has timed out once (or ``total_steps``), then report (num_success, total_dones) from the env's
        ``success_reset_buf``; a new best success rate is checkpointed as best_succ_*.  The reference's
        trajectory logging hooks into the same loop (``env.cfg_task.data_logger.collect_data``)."""
        logger = getattr(self, 'data_logger', None)
        save_trajectory = bool(self.env.cfg_task.data_logger.collect_data) and logger is not None
        self.set_eval()
        self.set_student_eval()
        obs_dict = self.env.reset(reset_at_success=False, reset_at_fails=False)
        steps, last = 0, int(self.env.max_episode_length) - 1
        while steps < min(total_steps, last):    # ext_adapt.py:616: stop when the episode clock runs out
            steps += 1
            mu, latent = self.student_act(obs_dict)
            obs_dict, r, done, info = self.env.step(mu)
            if save_trajectory:
                logger.log_trajectory_data(mu, latent, done, save_trajectory=save_trajectory)
        finished = self.env.test_reset_buf > 0   # envs whose episode has ended at least once
        num_success = int((self.env.success_reset_buf * finished).sum().item())
        total_dones = int(finished.sum().item())
        self.test_success = num_success / max(total_dones, 1)
        if self.output_dir is not None:                      # ext_adapt.py:623-629
            log_test_result(os.path.join(self.nn_dir, 'log.json'), best_loss=self.best_loss,
                            cur_loss=getattr(self, 'cur_loss', self.best_loss), best_reward=self.best_rewards,
                            cur_reward=self.cur_reward, steps=self.agent_steps, success_rate=self.test_success)
        best = getattr(self, 'best_success', -1.0)
        if self.output_dir is not None and self.test_success > best and self.agent_steps > 1e5:
            self.best_success = self.test_success
            self.save(os.path.join(self.nn_dir, f'best_succ_{self.best_success:.2f}'))
        return num_success, total_dones

    @torch.no_grad()
    def test_log(self, noise_levels=None, trials_per_noise=10, log_file=None):
        """ext_adapt.py:437-561: success rate of the student against Gaussian point-cloud noise (std 0 ... 1 cm),
        ``trials_per_noise`` episodes per level; returns {noise: {'mean', 'std'}} and writes it as JSON
        (the reference also draws a plot)."""
        import json
        if noise_levels is None:
            noise_levels = [0.01 * i / 9 for i in range(10)]
        self.set_eval()
        self.set_student_eval()
        results = {}
        for noise in noise_levels:
            rates = []
            for _ in range(trials_per_noise):
                obs_dict = self.env.reset(reset_at_success=False, reset_at_fails=False)
                for _ in range(int(self.env.max_episode_length)):
                    if obs_dict.get('pcl') is not None:
                        obs_dict['pcl'] = obs_dict['pcl'] + torch.randn_like(obs_dict['pcl']) * noise
                    mu, _ = self.student_act(obs_dict)
                    obs_dict, r, done, info = self.env.step(mu)
                rates.append(float(torch.mean(self.env.success_reset_buf * 1.0).item()))
            t = torch.tensor(rates)
            results[noise] = {'mean': float(t.mean()), 'std': float(t.std(unbiased=False))}
        if log_file is None and self.output_dir is not None:
            log_file = os.path.join(self.nn_dir, 'pcl_noise_success.json')
        if log_file is not None:
            with open(log_file, 'w') as f:
                json.dump({'results': {str(k): v for k, v in results.items()}}, f, indent=4)
        return results

    @torch.no_grad()
    def play_steps(self):
        """ext_adapt.py:658-767"""
        env_store = torch.ops.mi355ppo.rollout_env_store
        meter = torch.zeros((self.horizon_length, 4), dtype=torch.float32, device=self.device)
        dones_scratch = torch.empty(self.num_actors, dtype=torch.uint8, device=self.device)
        for n in range(self.horizon_length):
            self.it += 1
            n_obs = self.running_mean_std(self.obs['obs'])
            n_priv_info = self.priv_mean_std(self.obs['priv_info'])
            res_dict = self.agent.full_act({'obs': n_obs, 'priv_info': n_priv_info})
            student_dict = self.process_obs(self.obs)
            latent, _ = self.student.predict(student_dict, requires_grad=False)
            if not self.only_bc:                                   # ext_adapt.py:684-690
                student_actions, _ = self.agent.act_inference({'obs': n_obs, 'latent': latent})
            else:
                student_actions = latent
            if self.obs_info:
                self.storage.update_data('n_student_obs', n, student_dict['student_obs'])
            if self.seg_info:                                      # ext_adapt.py:695-698 (both under seg_info)
                if student_dict['img'] is not None:
                    self.storage.update_data('n_img', n, student_dict['img'])
                self.storage.update_data('n_seg', n, student_dict['seg'])
            elif self.img_info:
                self.storage.update_data('n_img', n, student_dict['img'])
            if self.tactile_info:
                self.storage.update_data('n_tactile', n, student_dict['tactile'])
            if self.pcl_info:
                self.storage.update_data('n_pcl', n, student_dict['pcl'].reshape(*self.env.pcl_queue.shape))
            self.storage.update_data('n_obs', n, n_obs)
            self.storage.update_data('n_priv_info', n, n_priv_info)
            self.storage.update_data('latent_gt', n, res_dict['latent_gt'])
            self.storage.update_data('teacher_actions', n, res_dict['actions'])
            self.storage.update_data('student_actions', n, student_actions)
            if self.agent_steps < 1e6 and not self.tactile_info:
                actions = torch.clamp(res_dict['actions'], -1.0, 1.0)
            else:                                                   # DAgger beta-mixing (:718-728)
                beta = max(0.0, 1.0 - self.agent_steps / 3e6)
                src = res_dict['actions'] if torch.rand(1).item() < beta else student_actions
                actions = torch.clamp(src, -1.0, 1.0)
            self.obs, rewards, self.dones, infos = self.env.step(actions)
            # dones / shaped reward / episode accumulators / meter sums in one launch, no host read-back
            # (ext_adapt.py:735-760 gathers the finished episodes with nonzero() every step)
            rewards = rewards.to(torch.float32).contiguous()
            dones = self.dones if self.dones.dtype == torch.uint8 else self.dones.to(torch.uint8)
            touts = infos.get('time_outs') if self.ppo_config['value_bootstrap'] else None
            if touts is not None:
                touts = (touts.view(torch.uint8) if touts.dtype == torch.bool else touts.to(torch.uint8)).contiguous()
            succ = infos['successes'].to(torch.float32).contiguous()
            values = res_dict['values'].to(torch.float32).contiguous()
            env_store(rewards, dones.contiguous(), values, touts, succ, float(self.ppo_config['gamma']),
                      touts is not None, self.storage.storage_dict['rewards'][n], dones_scratch, self.step_reward,
                      self.step_length, self.step_success, meter[n])
        self.mean_eps_reward.update_sums(meter[:, 0], meter[:, 3])
        self.mean_eps_length.update_sums(meter[:, 1], meter[:, 3])
        self.mean_eps_success.update_sums(meter[:, 2], meter[:, 3])
        self.agent_steps = (self.agent_steps + self.batch_size) if not self.multi_gpu \
            else self.agent_steps + self.batch_size * self.rank_size
        self.storage.prepare_training()

    def update_step(self, i):
        """Forward + loss + backward of minibatch ``i`` (ext_adapt.py:785-828): the raw local gradient is left in
        ``self.optim.flat_grad``.  Returns (action loss, latent loss)."""
        b = self.storage[i]
        if hasattr(b, 'prefetch'):           # the keys this step reads, gathered by one launch
            b.prefetch(('n_student_obs', 'n_tactile', 'n_img', 'n_seg', 'n_pcl', 'teacher_actions') +
                       (() if self.only_bc else ('n_obs', 'latent_gt')))
        student_dict = {
            'student_obs': b.get('n_student_obs'), 'tactile': b.get('n_tactile'), 'img': b.get('n_img'),
            'seg': b.get('n_seg'),
            'pcl': b['n_pcl'].reshape(b['n_pcl'].shape[0], -1, 3) if 'n_pcl' in b else None,
        }
        latent, _ = self.student.predict(student_dict, requires_grad=True)
        if not self.only_bc:                                 # act with the student latent (:799-806)
            mu, _ = self.agent.act_with_grad({'obs': b['n_obs'], 'latent': latent})
            loss_latent = torch.nn.functional.mse_loss(latent, b['latent_gt'].detach())
        else:                                                # pure behaviour cloning (:807-810)
            if getattr(self, '_zero1', None) is None:
                self._zero1 = torch.zeros(1, device=self.device)
            mu, loss_latent = latent, self._zero1
        # sum(w * (clamp(mu) - clamp(a_teacher))^2), a SUM, with d/dmu from the same kernel
        loss_action = bc_loss(mu, b['teacher_actions'], self.loss_weights)
        self.optim.zero_grad()
        # d(action_scale * loss): the scale enters as the root gradient -- the same value MulBackward hands down
        # (1.0 * scale), without the multiply, the ones-fill and the multiply of the backward pass
        if getattr(self, '_scale_v', None) != float(self.action_scale):       # (host-side compare: no device read)
            self._scale_v = float(self.action_scale)
            self._scale_t = torch.tensor(self._scale_v, dtype=torch.float32, device=self.device)
        loss_action.backward(gradient=self._scale_t)
        self.optim.sync_grads()              # the parameters' gradients -> the flat (all-reduce) buffer, one launch
        return loss_action.detach(), loss_latent.detach()

    def _native_comm(self):
        """The library's RCCL communicator of this rank (utils.dist.NativeComm), created on first use when the process
        group runs over RCCL; None under any other backend or with IGI_DP_NATIVE=0."""
        if not hasattr(self, "_comm"):
            from ...utils.dist import native_comm_or_none
            self._comm = native_comm_or_none(self.device, self.rank_size)
        return self._comm

    def update(self):
        """The optimisation half of train_epoch (ext_adapt.py:781-857) on the rollout in storage.

        Gradient exchange (:833-851 concatenates every gradient, all-reduces and copies back after backward): the flat
        gradient is reduced in place, SUM, with 1/world folded into the clip + Adam pass.  Over the library's own RCCL
        communicator it goes out in two buckets: the decoder side's range as soon as its gradients are final -- on the
        communication stream, under the backward of the tactile CNN / PointNets, which is most of the step -- and the
        encoders' range on the compute stream behind backward (``IGI_DP_OVERLAP=1``).  The DEFAULT for the student is the
        reference's single exchange after backward, through the same communicator: on a one-rank communicator the
        overlapped schedule costs +87 us per optimizer step (two collectives, three stream events, two gathers of the
        gradients instead of one: profiles/r04_dp_phase_cost.json) against the ~30 us of a 280 KB all-reduce it can hide,
        and the serial one costs nothing; ``bench.py --gpus N`` reports both.  Any other backend (gloo in the tests):
        one ``dist.all_reduce`` after backward."""
        latent_losses, action_losses = [], []
        comm = self._native_comm() if self.multi_gpu else None
        overlap = comm is not None and os.environ.get("IGI_DP_OVERLAP", "0") == "1"
        self.optim.arm_early(comm.all_reduce_async_ if overlap else None)
        for _ in range(self.mini_epochs_num):
            for i in range(len(self.storage)):
                loss_action, loss_latent = self.update_step(i)
                latent_losses.append(loss_latent)
                action_losses.append(loss_action)
                if self.grad_probe is not None:                      # raw (pre-reduce, pre-clip) gradient, for tests
                    self.grad_probe(len(action_losses) - 1, self.student.model)
                if overlap:                                          # the early range is in flight (or just went out)
                    late = self.optim.flat_grad[:self.optim.late_floats]
                    comm.join(self.device)          # first: never two collectives of one communicator on two streams
                    if late.numel():
                        comm.all_reduce_(late)
                elif comm is not None:
                    comm.all_reduce_(self.optim.grads())
                elif self.multi_gpu:                                 # :833-851 as one in-place collective
                    dist.all_reduce(self.optim.grads(), op=dist.ReduceOp.SUM)
                self.optim.step(1.0 / self.rank_size)                # clip 0.5 + Adam, 1/world folded in (:853-855)
        return action_losses, latent_losses

    def train_epoch(self):
        """ext_adapt.py:769-859"""
        self.set_student_eval()
        self.play_steps()
        self.set_student_train()
        return self.update()

    def update_student_alpha(self, steps, max_steps=1e6, init_alpha=0.01, final_alpha=1.0):
        """ext_adapt.py:377-381 (the blend weight is stored on the model; its use is commented out in tact.py:593)."""
        self.student.model.alpha = min(init_alpha + (final_alpha - init_alpha) * (steps / max_steps), 1.0)

    def _replace_best(self, prefix, old_value, new_value):
        prev = os.path.join(self.nn_dir, f'{prefix}_{old_value:.2f}.pth')
        for f in (prev, prev.replace('.pth', '_stud.pth')):
            if os.path.exists(f):
                os.remove(f)
        self.save(os.path.join(self.nn_dir, f'{prefix}_{new_value:.2f}'))

    def train(self):
        """ext_adapt.py:861-949: train_epoch until ``max_agent_steps``; every ``test_every`` (5e5) agent steps the
        student is evaluated without resets and ``stage2_nn/last{,_stud}.pth`` is written; best-loss / best-reward
        checkpoints replace their predecessors.  Multi-GPU statistics are averaged over ranks and logged by rank 0
        (the reference's multi-GPU branch leaves a_loss / l_loss unassigned: SURVEY Appendix A10)."""
        from ...utils.misc import multi_gpu_aggregate_stats
        _t = _last_t = time.time()
        test_every = getattr(self, 'test_every', 5e5)
        update_alpha, self.update_alpha_every = 1e4, 0
        self.epoch_num = 0
        self.next_test_step = test_every
        self.obs = self.env.reset(reset_at_success=True, reset_at_fails=True)
        self.agent_steps = self.batch_size if not self.multi_gpu else self.batch_size * self.rank_size
        if self.multi_gpu:
            dist.broadcast(self.optim.flat, 0)
        while self.agent_steps < self.max_agent_steps:
            self.epoch_num += 1
            a_losses, l_losses = self.train_epoch()
            a_loss, l_loss = torch.stack(a_losses).mean(), torch.stack(l_losses).mean()
            if self.multi_gpu:
                a_loss, l_loss = multi_gpu_aggregate_stats([a_loss.reshape(1), l_loss.reshape(1)])
                mean_rewards, mean_success = multi_gpu_aggregate_stats(
                    [torch.tensor([m.get_mean()], dtype=torch.float32, device=self.device)
                     for m in (self.mean_eps_reward, self.mean_eps_success)])
            else:
                a_loss, l_loss = a_loss.item(), l_loss.item()
                mean_rewards, mean_success = self.mean_eps_reward.get_mean(), self.mean_eps_success.get_mean()
            if not self.multi_gpu or self.rank == 0:
                now = time.time()
                print(f'ExtAdapt: Agent Steps: {int(self.agent_steps // 1e3):04}K | '
                      f'FPS: {self.agent_steps / (now - _t):.1f} | Last FPS: {self.batch_size / (now - _last_t):.1f} | '
                      f'Best Reward: {self.best_rewards:.2f} | Cur Reward: {mean_rewards:.2f} | '
                      f'Best Loss: {self.best_loss:.2f} | act_loss: {a_loss:.2f} | ext_loss: {l_loss:.2f}')
                _last_t = now
                self.cur_reward, self.cur_loss = mean_rewards, a_loss
                self.writer.add_scalar('losses/action_loss', a_loss, self.agent_steps)
                self.writer.add_scalar('losses/latent_loss', l_loss, self.agent_steps)
                self.writer.add_scalar('episode_rewards/step', mean_rewards, self.agent_steps)
                if self.agent_steps >= self.next_test_step:
                    self.test(total_steps=self.env.cfg_task.rl.max_episode_length)
                    self.obs = self.env.reset(reset_at_success=True, reset_at_fails=True)
                    self.set_student_train()
                    self.next_test_step += test_every
                    if self.output_dir is not None:
                        self.save(os.path.join(self.nn_dir, 'last'))
                if self.output_dir is not None and a_loss < self.best_loss and self.agent_steps > 1e5:
                    self._replace_best('best_loss', self.best_loss, a_loss)
                    self.best_loss = a_loss
                if self.output_dir is not None and mean_rewards > self.best_rewards and self.agent_steps >= 1e5 \
                        and mean_rewards != 0.0:
                    self._replace_best('best_reward', self.best_rewards, mean_rewards)
                    self.best_rewards = mean_rewards
                if self.tactile_info and self.agent_steps > self.update_alpha_every:
                    self.update_student_alpha(steps=self.agent_steps)
                    self.update_alpha_every += update_alpha
                self.success_rate = mean_success
        print('max steps achieved')

    # ------------------------------------------------------------------------------------------
    def save(self, name):
        """ext_adapt.py:1150-1170: {name}.pth = the (frozen) teacher with its normalisers, {name}_stud.pth = the
        student with its own; the running normalisers are left out for a student that keeps offline statistics."""
        torch.save({'model': self.agent.state_dict(), 'running_mean_std': self.running_mean_std.state_dict(),
                    'priv_mean_std': self.priv_mean_std.state_dict()}, f'{name}.pth')
        weights = {'student': self.student.model.state_dict()}
        if not self.train_config.from_offline:
            weights['stud_obs_mean_std'] = self.stud_obs_mean_std.state_dict()
            weights['pcl_mean_std'] = self.pcl_mean_std.state_dict()
        torch.save(weights, f'{name}_stud.pth')

    def restore_train(self, fn, restore_student=False, phase=None):
        """ext_adapt.py:1074-1084: the teacher checkpoint and, optionally, the student saved next to it
        (``stage1_nn/last.pth`` -> ``stage2_nn/last_stud.pth``; any other name -> ``<name>_stud.pth``)."""
        checkpoint = torch.load(fn, map_location=self.device)
        self.agent.load_state_dict(checkpoint['model'])
        self.running_mean_std.load_state_dict(checkpoint['running_mean_std'])
        self.priv_mean_std.load_state_dict(checkpoint['priv_mean_std'])
        self.set_eval()
        if restore_student:
            stud_fn = fn.replace('stage1_nn/last.pth', 'stage2_nn/last_stud.pth')
            if stud_fn == fn:
                stud_fn = fn.replace('.pth', '_stud.pth')
            self.restore_student(stud_fn, from_offline=self.train_config.from_offline, phase=phase)

    def restore_student(self, fn, from_offline=False, phase=None):
        """ext_adapt.py:1099-1135.  ``from_offline``: the student comes from the offline supervised run
        (``checkpoints/model_last.pt`` = a bare state_dict, or a stage-2 ``*_stud.pth`` when ``phase == 2``) and
        its proprioception keeps the dataset statistics of ``normalization.pkl``; otherwise the online checkpoint
        with its running normalisers (non-strict model load, as in the reference)."""
        if from_offline:
            import pickle
            if phase == 2:
                self.student.model.load_state_dict(torch.load(fn, map_location=self.device)['student'])
            else:
                self.student.model.load_state_dict(
                    torch.load(self.train_config.train.student_ckpt_path, map_location=self.device))
            with open(self.train_config.train.normalize_file, "rb") as f:
                stats = pickle.load(f)
            self.stats = {kind: {k: torch.as_tensor(v, dtype=torch.float32, device=self.device)
                                 for k, v in stats[kind].items()} for kind in ('mean', 'std')}
            return
        self.stats = None
        checkpoint = torch.load(fn, map_location=self.device)
        self.stud_obs_mean_std.load_state_dict(checkpoint['stud_obs_mean_std'])
        if 'pcl_mean_std' in checkpoint:
            self.pcl_mean_std.load_state_dict(checkpoint['pcl_mean_std'])
        self.student.model.load_state_dict(checkpoint['student'], strict=False)
        if phase == 3:
            # ext_adapt.py:1136-1147: only the tactile branch and layers named 'new' keep training, with their own
            # optimizer Adam(lr=1e-3, weight_decay=1e-6) -- torch's coupled L2 decay
            chosen = []
            for name, p in self.student.model.named_parameters():
                if 'tac' in name or 'new' in name:
                    chosen.append(p)
                else:
                    p.requires_grad = False
                    p.grad = None
            if not chosen:
                raise RuntimeError("phase 3 trains the tactile / 'new' layers, but the student has none")
            self.optim = FlatAdam(chosen, lr=1e-3, max_norm=0.5, l2=1e-6)

    def restore_test(self, fn):
        """ext_adapt.py:1086-1097"""
        self.restore_train(fn, restore_student=True, phase=1)
        self.set_eval()
        self.set_student_eval()
