"""Stage-1 (privileged teacher) deployment player with the reference's names (algo/deploy/deploy_s1.py:26-133,
708-786): ``HardwarePlayer(full_config)``, ``restore(fn)``, ``set_eval()``, ``deploy()``.

The networks are the reference's (``ActorCriticSplit`` + three ``RunningMeanStd``), built from the same config
entries, and ``restore`` consumes the same ``stage1_nn/*.pth`` keys, so a checkpoint written by either code base
loads in the other.  One control tick = normalise obs / priv_info (eval-mode statistics), ``act_inference``,
clamp to [-1, 1] (deploy_s1.py:757-768) -- on the device: igi_rms_forward + igi_teacher_infer."""
import torch

from ..models.models_split import ActorCriticSplit as ActorCritic
from ..models.running_mean_std import RunningMeanStd


class HardwarePlayer:
    def __init__(self, full_config, robot=None):
        self.num_envs = 1
        self.full_config = full_config
        self.deploy_config = full_config.get('deploy', None)
        self.device = full_config["rl_device"]
        env, ppo, net = full_config.task.env, full_config.train.ppo, full_config.train.network
        self.num_observations = env.numObservations
        self.num_obs_stud = env.numObsStudent
        self.obs_shape = (env.numObservations,)
        self.obs_stud_shape = (env.numObsStudent,)
        self.num_actions = env.numActions
        self.priv_info = ppo.priv_info
        self.priv_info_dim = ppo.priv_info_dim
        self.obs_info = ppo.obs_info
        self.student_obs_input_shape = ppo.get('student_obs_input_shape', env.numObsStudent)
        self.gt_contacts_info = ppo.compute_contact_gt
        net_config = {
            'actor_units': net.mlp.units, 'actions_num': self.num_actions, 'priv_mlp_units': net.priv_mlp.units,
            'input_shape': self.obs_shape, 'priv_info_dim': self.priv_info_dim, 'priv_info': self.priv_info,
            'obs_info': self.obs_info, 'gt_contacts_info': self.gt_contacts_info,
            'only_contact': ppo.only_contact, 'contacts_mlp_units': net.contact_mlp.units,
            'shared_parameters': False,
        }
        self.model = ActorCritic(net_config)
        self.model.to(self.device)
        self.model.eval()
        self.running_mean_std = RunningMeanStd(self.obs_shape).to(self.device)
        self.running_mean_std_stud = RunningMeanStd((self.student_obs_input_shape,)).to(self.device)
        self.priv_mean_std = RunningMeanStd((self.priv_info_dim,)).to(self.device)
        self.set_eval()
        self.env = robot
        self.episode_length = torch.zeros((1, 1), device=self.device, dtype=torch.float)

    def restore(self, fn):
        """deploy_s1.py:114-127"""
        checkpoint = torch.load(fn, map_location=self.device)
        self.running_mean_std.load_state_dict(checkpoint['running_mean_std'])
        if 'running_mean_std_stud' in checkpoint:
            self.running_mean_std_stud.load_state_dict(checkpoint['running_mean_std_stud'])
        if 'priv_mean_std' in checkpoint:
            self.priv_mean_std.load_state_dict(checkpoint['priv_mean_std'])
        self.model.load_state_dict(checkpoint['model'])
        self.set_eval()

    def set_eval(self):
        """deploy_s1.py:129-133"""
        self.model.eval()
        self.running_mean_std.eval()
        self.running_mean_std_stud.eval()
        self.priv_mean_std.eval()

    @torch.no_grad()
    def policy_step(self, obs, priv):
        """deploy_s1.py:757-768: (action clamped to [-1, 1], latent)."""
        input_dict = {'obs': self.running_mean_std(obs.clone()), 'priv_info': self.priv_mean_std(priv.clone())}
        action, latent = self.model.act_inference(input_dict)
        return torch.clamp(action, -1.0, 1.0), latent

    def deploy(self, num_episodes=5):
        """deploy_s1.py:708-786 with the robot behind RobotIO; returns the number of control ticks."""
        if self.env is None:
            raise RuntimeError("HardwarePlayer.deploy needs a RobotIO (robot=...): the ROS / MoveIt side of the "
                               "reference's player is not part of this package")
        ticks, cur_episode = 0, 0
        while cur_episode < num_episodes:
            self.episode_length.zero_()
            while True:
                o = self.env.observe()
                action, _ = self.policy_step(o['obs'].to(self.device), o['priv_info'].to(self.device))
                self.env.apply(action)
                self.episode_length += 1
                ticks += 1
                if self.env.done():
                    cur_episode += 1
                    break
            self.env.reset()
        return ticks
