"""Student wrapper with the reference's interface (algo/models/transformer/runner.py:25-148, 374-441):
``Runner(cfg, agent=None, action_regularization=False)`` builds the ``MultiModalModel`` from
``cfg.offline_train.model.*`` + ``cfg.task.env`` point-cloud counts; ``predict(obs_dict,
requires_grad)`` reshapes the tactile queue (B,T,3,H*W) -> (B,T,3,1,H,W) and runs the model.

The reference's eval tactile transform (Resize + CenterCrop to the same size, applied per image in a
Python loop, utils.py:131-156) is the identity at the configured sizes and is skipped
(SURVEY section 8 a-12).  The offline supervised loop (run / run_train) is the next scope row (8f-2).
"""
import torch

from .tact import MultiModalModel


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
        """runner.py:150-192 (sizes only; the eval transforms are identities at these sizes)."""
        self.num_fingers = 3
        self.tactile_channel = 1 if self.cfg.tactile_type == "gray" else 3
        self.tactile_width = self.cfg.tactile_width
        self.tactile_height = self.cfg.tactile_height
        self.crop_tactile_width = self.tactile_width - self.cfg.get('tactile_crop_w', 0)
        self.crop_tactile_height = self.tactile_height - self.cfg.get('tactile_crop_h', 0)
        self.tactile_transform = True
        self.eval_process_tactile = lambda t: t

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
        if self.cfg.model.use_lin:
            student_obs = student_obs.to(self.device)
        if self.cfg.model.use_pcl:
            pcl = pcl.to(self.device)
        out = self.model(obs_tactile=tactile, obs_img=None, obs_seg=None, lin_input=student_obs, obs_pcl=pcl)
        return out, None
