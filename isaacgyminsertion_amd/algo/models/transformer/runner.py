"""Student wrapper with the reference's interface (algo/models/transformer/runner.py:25-655):
``Runner(cfg, agent=None, action_regularization=False)`` builds the ``MultiModalModel`` from
``cfg.offline_train.model.*`` + ``cfg.task.env`` point-cloud counts; ``predict(obs_dict,
requires_grad)`` reshapes the tactile queue (B,T,3,H*W) -> (B,T,3,1,H,W) and runs the model;
``run()`` is the offline supervised loop of ``train_supervised.py`` (BASELINE configs[0]): glob the
trajectory files, ``DataNormalizer``, 98/2 split, ``AdamW(lr, weight_decay=1e-6)``, per batch
``MSE(out, action[:, -1])`` -> clip_grad_norm_(0.5) -> step (runner.py:194-304, 470-576).

MI355X-side differences (same arithmetic):
  * the optimiser step is ONE native launch pair (igi_clip_adamw: global-norm clip + AdamW over the flat
    parameter vector); every Linear is igi_linear_forward/backward;
  * batches come from ``ResidentLoader`` (all sub-sequences resident in HBM, shuffled by an index
    gather) instead of 16 DataLoader workers; any iterable of the reference's 8-tuples is accepted;
  * losses stay on the device and are read back once per ``print_every`` window, not per step
    (the reference calls ``.item()`` three times per batch, runner.py:250-253);
  * the eval tactile transform (Resize + CenterCrop to the same size, applied per image in a Python loop,
    utils.py:131-156) is one batched op, the identity at the configured sizes.
Figures / wandb / ``log_output`` are not produced (outside the scope table).
"""
import os
import random
from datetime import datetime
from glob import glob

import numpy as np
import torch

from ....optim import FlatAdam
from .data import DataNormalizer, ResidentLoader, TactileDataset
from .tact import MultiModalModel
from .utils import TactileTransform, define_tactile_transforms, log_output


class _Schedule:
    """Epoch-level learning-rate schedules of run_train (runner.py:483-500) for the flat optimiser:
    'cosine' = CosineAnnealingLR(T_max=epochs) in closed form, 'reduce' = ReduceLROnPlateau(min, 0.5,
    patience 3, rel threshold 1e-4), optional linear warm-up (GradualWarmupScheduler, multiplier 1)."""

    def __init__(self, opt, kind, epochs, warmup_epochs=0):
        self.opt, self.kind, self.epochs, self.warmup = opt, kind, max(int(epochs), 1), int(warmup_epochs)
        self.base = opt.param_groups[0]["lr"]
        self.epoch, self.best, self.bad, self.scale = 0, float("inf"), 0, 1.0
        if self.warmup > 0:
            self.opt.param_groups[0]["lr"] = 0.0

    def step(self, metric=None):
        self.epoch += 1
        if self.warmup > 0 and self.epoch <= self.warmup:
            lr = self.base * self.epoch / self.warmup
        else:
            e = self.epoch - self.warmup
            if self.kind == 'cosine':
                lr = self.base * (1 + np.cos(np.pi * e / self.epochs)) / 2
            elif self.kind == 'reduce':
                if metric is not None and metric < self.best * (1 - 1e-4):
                    self.best, self.bad = metric, 0
                else:
                    self.bad += 1
                if self.bad > 3:
                    self.scale *= 0.5
                    self.bad = 0
                lr = self.base * self.scale
            else:
                lr = self.base
        self.opt.param_groups[0]["lr"] = float(lr)


class Runner:
    def __init__(self, cfg=None, agent=None, action_regularization=False):
        self.task_cfg = cfg
        self.cfg = cfg.offline_train
        self.agent = agent
        self.only_bc = self.cfg.only_bc
        self.ppo_step = agent.play_latent_step if ((agent is not None) and action_regularization) else None
        self.optimizer = None
        self.scheduler = None
        self.tact = None
        self.sequence_length = self.cfg.model.transformer.sequence_length
        gpu = self.cfg.gpu_ids[0] if 'gpu_ids' in self.cfg else 0
        self.device = cfg.get('rl_device', f'cuda:{gpu}') if hasattr(cfg, 'get') else f'cuda:{gpu}'
        self._init_transforms()
        self.init_model()

    def _init_transforms(self):
        """runner.py:150-192, tactile part (img / seg transforms belong to the depth branch)."""
        self.num_fingers = 3
        self.tactile_channel = 1 if self.cfg.tactile_type == "gray" else 3
        self.tactile_color_jitter = self.cfg.get('tactile_color_jitter', False)
        self.tactile_width = self.cfg.tactile_width
        self.tactile_height = self.cfg.tactile_height
        self.crop_tactile_width = self.tactile_width - self.cfg.get('tactile_crop_w', 0)
        self.crop_tactile_height = self.tactile_height - self.cfg.get('tactile_crop_h', 0)
        self.tactile_transform, self.tactile_eval_transform = define_tactile_transforms(
            self.tactile_width, self.tactile_height, self.crop_tactile_width, self.crop_tactile_height,
            self.cfg.get('tactile_patch_size', 16), self.cfg.get('tactile_gaussian_noise', 0.0),
            self.cfg.get('tactile_masking_prob', 0.0))
        self.process_tactile = TactileTransform(self.tactile_transform)
        self.eval_process_tactile = TactileTransform(self.tactile_eval_transform)
        # external camera (runner.py:152-173): sizes only; the eval pipeline (centre crop to the crop size, resize
        # to (img_width, img_height), centre crop) is the identity when nothing is cropped
        self.img_channel = 1 if self.cfg.get('img_type', 'depth') == "depth" else 3
        self.img_width, self.img_height = self.cfg.get('img_width', 54), self.cfg.get('img_height', 96)
        self.crop_img_width = self.img_width - self.cfg.get('img_crop_w', 0)
        self.crop_img_height = self.img_height - self.cfg.get('img_crop_h', 0)
        self.img_transform = self.seg_transform = self.sync_transform = None
        self.img_eval_transform = self.sync_eval_transform = None

    def init_model(self):
        """runner.py:78-148 (model_type 'tact')."""
        out_size = 6 if self.only_bc else self.cfg.model.transformer.output_size
        if self.cfg.model.model_type != 'tact':
            raise NotImplementedError("only model_type='tact' is on the hot path (offline_config.yaml:92)")
        env = self.task_cfg.task.env
        pcl_conf = {'num_sample_plug': env.num_points, 'num_sample_hole': env.num_points_socket,
                    'num_sample_goal': env.num_points_goal, 'num_sample_all': env.num_points_goal,
                    'merge_socket': env.merge_socket_pcl, 'merge_goal': env.merge_goal_pcl,
                    'scene_pcl': env.include_all_pcl, 'merge_plug': env.include_plug_pcl, 'relative': False}
        tr = self.cfg.model.transformer
        self.model = MultiModalModel(
            context_size=self.sequence_length, num_channels=self.tactile_channel,
            num_lin_features=self.cfg.model.linear.input_size, num_outputs=out_size, tactile_encoder="depth",
            img_encoder="depth", seg_encoder="depth", lin_encoding_size=tr.lin_encoding_size,
            tactile_encoding_size=tr.tactile_encoding_size, img_encoding_size=tr.img_encoding_size,
            seg_encoding_size=tr.seg_encoding_size, mha_num_attention_heads=tr.num_heads,
            mha_num_attention_layers=tr.num_layers, mha_ff_dim_factor=tr.dim_factor, additional_lin=0,
            include_img=self.cfg.model.use_img, include_seg=self.cfg.model.use_seg,
            include_lin=self.cfg.model.use_lin, include_pcl=self.cfg.model.use_pcl,
            include_tactile=self.cfg.model.use_tactile, only_bc=self.only_bc, pcl_conf=pcl_conf)
        self.model.to(self.device)
        return self.model

    def predict(self, obs_dict, requires_grad=False, display=False):
        """runner.py:374-381"""
        if not requires_grad:
            self.model.eval()
            with torch.no_grad():
                return self._predict_forward(obs_dict, display)
        return self._predict_forward(obs_dict, display)

    def _predict_forward(self, obs_dict, display=False):
        """runner.py:383-441"""
        tactile = obs_dict.get('tactile')
        student_obs = obs_dict.get('student_obs')
        pcl = obs_dict.get('pcl')
        if self.cfg.model.use_tactile:
            tactile = tactile.to(self.device)
            if tactile.ndim == 4:      # (B, T, fingers, C*H*W) -> (B, T, F, C, W, H) as the reference names them
                tactile = tactile.reshape(*tactile.shape[:2], self.num_fingers, 1, self.crop_tactile_width,
                                          self.crop_tactile_height)
        img, seg = obs_dict.get('img'), obs_dict.get('seg')
        if self.cfg.model.use_img or self.cfg.model.use_seg:
            if (self.crop_img_width, self.crop_img_height) != (self.img_width, self.img_height):
                raise NotImplementedError("cropped camera images (img_crop_w/h > 0) are not built")
            shp = (1, self.crop_img_width, self.crop_img_height)        # (B, T, H*W) -> (B, T, C, W, H), runner.py:419-424
            img = img.to(self.device).reshape(*img.shape[:2], *shp) if img is not None else None
            seg = seg.to(self.device).reshape(*seg.shape[:2], *shp) if seg is not None else None
        if self.cfg.model.use_lin:
            student_obs = student_obs.to(self.device)
        if self.cfg.model.use_pcl:
            pcl = pcl.to(self.device)
        out = self.model(obs_tactile=tactile, obs_img=img, obs_seg=seg, lin_input=student_obs, obs_pcl=pcl)
        return out, None

    # ------------------------------------------------------------------------------------------
    # offline supervised training (train_supervised.py:40-45 -> run)
    # ------------------------------------------------------------------------------------------
    def _forward_loss(self, batch, clamp):
        tactile, img, seg, stud_obs, pos_rpy, obs_hist, latent, action = batch
        dev = self.device
        tactile = tactile.to(dev) if self.cfg.model.use_tactile else None
        img = img.to(dev) if self.cfg.model.use_img else None
        seg = seg.to(dev) if self.cfg.model.use_seg else None
        stud_obs, latent, action = stud_obs.to(dev), latent.to(dev), action.to(dev)
        out = self.model(tactile, img, seg, stud_obs, add_lin_input=None)
        if self.only_bc:
            if clamp:
                out = torch.clamp(out, -1, 1)                     # validation only (runner.py:326)
            loss_latent = self.loss_fn_mean(out, action[:, -1, :])
        else:
            loss_latent = self.loss_fn_mean(out, latent[:, -1, :])
        loss_action = torch.zeros(1, device=dev)
        if self.ppo_step is not None:
            oh = obs_hist[:, -1, :].to(dev).view(obs_hist.shape[0], obs_hist.shape[-1])
            pred_action, _ = self.ppo_step({'obs': oh, 'latent': out})
            if clamp:
                pred_action = torch.clamp(pred_action, -1, 1)
            loss_action = self.loss_fn_mean(pred_action, action[:, -1, :])
        loss = (self.cfg.train.latent_scale * loss_latent) + (self.cfg.train.action_scale * loss_action)
        return loss.reshape(()), loss_latent, loss_action, out

    def train(self, dl, val_dl, ckpt_path, print_every=50, eval_every=250, test_every=500):
        """runner.py:194-304: one pass over ``dl``; returns the last validation loss."""
        self.model.train()
        window, val_loss = [], []
        n_batches = len(dl)
        for i, batch in enumerate(dl):
            self.model.train()
            loss, loss_latent, loss_action, out = self._forward_loss(batch, clamp=False)
            self.optimizer.zero_grad()
            loss.backward()
            self.optimizer.step()                                 # clip_grad_norm_(0.5) + AdamW, fused
            window.append(loss.detach())
            last = (i == n_batches - 1)
            if (i + 1) % print_every == 0 or last:
                mean = float(torch.stack(window).mean())          # one read-back per window
                print(f'step {i + 1}:', mean)
                self._wandb_log({'train/loss': mean})
                self.train_loss.append(mean)
                window = []
            if (i + 1) % eval_every == 0 or last:
                val_loss = self.validate(val_dl)
                print(f'validation loss: {val_loss}')
                self.val_loss.append(val_loss)
                log_output()
                self.model.train()
        return val_loss

    def validate(self, dl):
        """runner.py:306-372: mean of the per-batch losses, outputs clamped to [-1, 1] under only_bc."""
        self.model.eval()
        losses = []
        with torch.no_grad():
            for batch in dl:
                loss, _, _, _ = self._forward_loss(batch, clamp=True)
                losses.append(loss)
        if not losses:
            return float('nan')
        return float(torch.stack(losses).mean())

    def test(self):
        """runner.py:443-455"""
        with torch.inference_mode():
            num_success, total_trials = self.agent.test(self.predict, self.stats.copy())
            if total_trials > 0:
                print(f'{num_success}/{total_trials}, success rate on :', num_success / total_trials)
                self._wandb_log({'test/success_rate': num_success / total_trials})
            else:
                print('something went wrong, there are no test trials')

    def load_model(self, model_path, device='cuda:0'):
        """runner.py:457-462"""
        print('Loading Multimodal model:', model_path)
        self.model.load_state_dict(torch.load(model_path, map_location=device))
        self.device = device
        self.model.to(device)

    def _make_optimizer(self, learning_rate):
        """runner.py:481 ``AdamW(model.parameters(), lr, weight_decay=1e-6)``.  torch skips parameters whose
        ``.grad`` is None -- here that is exactly ``decoder.sa_layer.*``, the registered-but-never-called
        encoder-layer template -- so those stay out of the flat vector and are
        neither decayed nor stepped."""
        params = [p for n, p in self.model.named_parameters() if not n.startswith('decoder.sa_layer.')]
        return FlatAdam(params, lr=learning_rate, max_norm=0.5, weight_decay=1e-6)

    def _make_loader(self, files, batch_size, train):
        ds = TactileDataset(traj_files=files, sequence_length=self.sequence_length, stats=self.stats,
                            tactile_transform=None, include_img=self.cfg.model.use_img,
                            include_seg=self.cfg.model.use_seg, include_lin=self.cfg.model.use_lin,
                            include_tactile=self.cfg.model.use_tactile, obs_keys=self.cfg.train.obs_keys)
        tf = (self.tactile_transform if train else self.tactile_eval_transform) if self.cfg.model.use_tactile else None
        if tf is not None:
            tf.to(self.device)
        return ResidentLoader(ds, batch_size, shuffle=True, device=self.device, tactile_transform=tf)

    def run_train(self, file_list, save_folder, epochs=100, train_test_split=0.9, train_batch_size=32,
                  val_batch_size=32, learning_rate=1e-4, device='cuda:0', print_every=50, eval_every=250,
                  test_every=500):
        """runner.py:470-576"""
        random.shuffle(file_list)
        print('# trajectories:', len(file_list))
        ckpt_path = f'{save_folder}/checkpoints'
        os.makedirs(ckpt_path, exist_ok=True)
        self.optimizer = self._make_optimizer(learning_rate)
        kind = self.cfg.train.get('scheduler', None)
        warm = self.cfg.train.get('warmup_epochs', 0) if self.cfg.train.get('warmup', False) else 0
        self.scheduler = _Schedule(self.optimizer, kind, epochs, warm) if (kind in ('cosine', 'reduce') or warm) else None
        n_train = int(len(file_list) * train_test_split)
        train_dl = self._make_loader(file_list[:n_train], train_batch_size, True)
        val_dl = self._make_loader(file_list[n_train:], val_batch_size, False)
        for epoch in range(epochs):
            self.validate(val_dl)
            if self.cfg.train.get('only_validate', False):
                self.validate(val_dl)
                continue
            val_loss = self.train(train_dl, val_dl, ckpt_path, print_every=print_every, eval_every=eval_every,
                                  test_every=test_every)
            if self.scheduler is not None:
                self.scheduler.step(float(np.mean(val_loss)))
            print('Saving the model')
            torch.save(self.model.state_dict(), f'{ckpt_path}/model_last.pt')

    def _wandb_log(self, data):
        if self.cfg.get('wandb', None) is not None and self.cfg.wandb.get('wandb_enabled', False):
            raise NotImplementedError("wandb is not available in this environment; set wandb_enabled: False")

    def run(self):
        """runner.py:578-641"""
        self.loss_fn_mean = torch.nn.MSELoss(reduction='mean')
        self.loss_fn = torch.nn.MSELoss(reduction='none')
        self.train_loss, self.val_loss = [], []
        if self.cfg.train.get('load_checkpoint', False):
            self.load_model(self.cfg.train.student_ckpt_path, device=self.device)
        train_config = {k: self.cfg.train[k] for k in ("epochs", "train_test_split", "train_batch_size",
                                                        "val_batch_size", "learning_rate", "print_every",
                                                        "eval_every", "test_every")}
        print('Loading trajectories from', self.cfg.data_folder)
        file_list = glob(os.path.join(self.cfg.data_folder, '*/*/obs/*.npz'))
        save_folder = os.path.join(os.path.abspath(self.cfg.output_dir),
                                   f'{self.cfg.model.model_type}_{datetime.now().strftime("%Y-%m-%d_%H-%M-%S")}')
        os.makedirs(save_folder, exist_ok=True)
        self.save_folder = save_folder
        normalizer = DataNormalizer(self.cfg, file_list, self.cfg.data_folder)
        normalizer.run()
        self.stats = normalizer.stats
        file_list = normalizer.file_list
        if self.cfg.train.get('only_test', False):
            print('Only testing')
            self.test()
        self.model = self.model.to(self.device)
        self.run_train(file_list, save_folder, device=self.device, **train_config)
