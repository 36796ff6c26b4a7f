"""Synthetic stand-in for the IsaacGym VecTask (closed-source, CUDA-only): honours the
observation-dict / step contract the trainers consume (vec_task.py:302-326, 401-415;
factory_task_insertion.py:2126-2175) with seeded random tensors generated on the device, so the
learning side runs unchanged without a simulator.  Recording hooks are no-ops."""
import torch

from ..utils.config import AttrDict, to_attr


class SyntheticInsertionEnv:
    def __init__(self, num_envs=4096, obs_dim=15, priv_dim=64, act_dim=6, device="cuda:0", seed=1234,
                 done_p=0.01, reward_scale=0.1, max_episode_length=512, tactile_hw=None, pcl_points=0, img_hw=None):
        self.num_envs, self.obs_dim, self.priv_dim, self.act_dim = num_envs, obs_dim, priv_dim, act_dim
        self.device = torch.device(device)
        self.gen = torch.Generator(device=self.device).manual_seed(seed)
        self.done_p, self.reward_scale = done_p, reward_scale
        self.cfg_task = to_attr({"data_logger": {"collect_data": False},
                                 "rl": {"max_episode_length": max_episode_length},
                                 "env": {"record_video_every": 10 ** 9, "record_ft_every": 10 ** 9},
                                 "external_cam": {"display": False}})
        self.max_episode_length = max_episode_length
        self.progress = torch.zeros(num_envs, device=self.device)
        # evaluation bookkeeping read by the trainers' test loops (factory_task_insertion.py: success_reset_buf =
        # "this env's last episode ended inserted", test_reset_buf = "has finished once since reset"); a synthetic
        # episode "succeeds" with probability success_p when it ends
        self.success_p = 0.5
        self.gen_eval = torch.Generator(device=self.device).manual_seed(seed + 1)   # keeps the obs stream as is
        self.success_reset_buf = torch.zeros(num_envs, dtype=torch.long, device=self.device)
        self.test_reset_buf = torch.zeros(num_envs, dtype=torch.long, device=self.device)
        # student modalities (factory_task_insertion.py:305-345): tactile queue (N, hist=1, 3 fingers, C*H*W)
        # of gray crops, point-cloud queue (N, hist=1, points*3) = plug points then socket points
        self.tactile_hw, self.pcl_points = tactile_hw, pcl_points
        self.tactile_queue = torch.zeros(num_envs, 1, 3, tactile_hw[0] * tactile_hw[1], device=self.device) \
            if tactile_hw else None
        self.pcl_queue = torch.zeros(num_envs, 1, pcl_points * 3, device=self.device) if pcl_points else None
        # external camera (factory_task_insertion.py: image_buf / seg_buf queues): depth (N, hist=1, H*W) in [0, 1]
        # and the segmentation ids of the same pixels (0 background, 1 robot, 2 plug, 3 socket)
        self.img_hw = img_hw
        self.img_queue = torch.zeros(num_envs, 1, img_hw[0] * img_hw[1], device=self.device) if img_hw else None
        self.seg_queue = torch.zeros(num_envs, 1, img_hw[0] * img_hw[1], device=self.device) if img_hw else None

    def _obs(self):
        n, d = self.num_envs, self.device
        o = {"obs": torch.randn(n, self.obs_dim, generator=self.gen, device=d),
             "priv_info": torch.randn(n, self.priv_dim, generator=self.gen, device=d),
             "student_obs": torch.randn(n, self.obs_dim, generator=self.gen, device=d)}
        if self.tactile_hw:
            o["tactile"] = torch.rand(self.tactile_queue.shape, generator=self.gen, device=d)
        if self.img_hw:
            o["img"] = torch.rand(self.img_queue.shape, generator=self.gen, device=d)
            o["seg"] = torch.randint(0, 4, self.seg_queue.shape, generator=self.gen, device=d).float()
        if self.pcl_points:
            c = 0.3 * torch.randn(n, 1, 1, 3, generator=self.gen, device=d)
            p = c + 0.05 * torch.randn(n, 1, self.pcl_points, 3, generator=self.gen, device=d)
            o["pcl"] = p.reshape(n, 1, self.pcl_points * 3)
        return o

    @property
    def progress_buf(self):
        return self.progress.long()

    def reset(self, reset_at_success=False, reset_at_fails=True):
        self.progress.zero_()
        self.success_reset_buf.zero_()
        self.test_reset_buf.zero_()
        return self._obs()

    def step(self, actions):
        assert actions.shape == (self.num_envs, self.act_dim)
        n, d = self.num_envs, self.device
        rewards = self.reward_scale * torch.randn(n, generator=self.gen, device=d)
        dones = (torch.rand(n, generator=self.gen, device=d) < self.done_p)
        self.progress += 1
        time_outs = self.progress >= self.cfg_task.rl.max_episode_length - 1   # factory_task_insertion.py:1189
        dones = dones | time_outs
        self.progress = self.progress * (~dones)
        success = dones & (torch.rand(n, generator=self.gen_eval, device=d) < self.success_p)
        self.success_reset_buf = torch.where(dones, success.long(), self.success_reset_buf)
        self.test_reset_buf = self.test_reset_buf | dones.long()
        infos = {"time_outs": time_outs, "successes": success.float()}
        return self._obs(), rewards, dones.to(torch.uint8), infos

    # video / force-plot hooks used by PPO.log_video (frozen_ppo.py:791-851)
    def start_recording(self): pass
    def stop_recording(self): pass
    def pause_recording(self): pass
    def start_recording_ft(self): pass
    def stop_recording_ft(self): pass
