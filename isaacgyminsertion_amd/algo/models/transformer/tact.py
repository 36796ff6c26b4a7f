"""Student network with the reference's interface and state_dict
(algo/models/transformer/tact.py:115-212, 214-599): ``PositionalEncoding``, ``MultiLayerDecoder``,
``MLPDecoder``, ``MultiModalModel``.

Tokens are built in the reference's order [tactile, img, seg, lin, pcl], each (B, 1, 32):
  * tactile -> CNNWithSpatialSoftArgmax  : HIP implicit-GEMM convolutions + soft-argmax (tactile_cnn.py)
  * pcl     -> PointNet x {plug, socket} : HIP MFMA + running arg-max (pointnets.py) -> compress MLP
  * lin     -> Linear(15,64)-ReLU-Linear(64,32)
and decoded by the 2-layer, 3-token, d=32 transformer (or the MLP decoder when no tactile token is
present) and the Linear(32, 6)+Tanh head.  Every Linear outside the attention blocks (lin encoder,
point-cloud compress, decoder output stack, MLP decoder, head) is a ``HipLinear`` = igi_linear_forward /
igi_linear_backward with the following ReLU / Tanh fused into the GEMM epilogue, and the 2-layer token
transformer is ``HipTransformerEncoder`` = igi_token_forward / igi_token_backward.  Only tensor plumbing
(token concatenation, the positional-encoding add, reshapes) is left to PyTorch.
  * img / seg -> DepthOnlyFCBackbone54x96 : HIP MFMA conv + pool, implicit-GEMM conv, split-K Linear (depth_backbone.py)
The efficientnet encoders are outside the scope table and raise.
"""
import math
import os
from typing import Dict, Optional

import torch
import torch.nn as nn

from ....hip_linear import HipLinear, mlp_chain, run_chain
from ....hip_token_encoder import HipTransformerEncoder
from .depth_backbone import DepthOnlyFCBackbone54x96
from .pointnets import PointNet

# IGI_PCL_ONE_LAUNCH=0: one PointNet launch per object + a concatenation (A/B; read once)
_PCL_ONE_LAUNCH = os.environ.get("IGI_PCL_ONE_LAUNCH", "1") != "0"
from .tactile_cnn import CNNWithSpatialSoftArgmax


class PositionalEncoding(nn.Module):
    """tact.py:115-134"""

    def __init__(self, d_model, max_seq_len=6):
        super().__init__()
        pos_enc = torch.zeros(max_seq_len, d_model)
        pos = torch.arange(0, max_seq_len, dtype=torch.float).unsqueeze(1)
        div_term = torch.exp(torch.arange(0, d_model, 2).float() * (-math.log(10000.0) / d_model))
        pos_enc[:, 0::2] = torch.sin(pos * div_term)
        pos_enc[:, 1::2] = torch.cos(pos * div_term)
        self.register_buffer('pos_enc', pos_enc.unsqueeze(0))

    def forward(self, x):
        return x + self.pos_enc[:, :x.size(1), :]


class MultiLayerDecoder(nn.Module):
    """tact.py:137-158.  ``sa_layer`` stays a registered (never trained) template, as in the reference
    (SURVEY Appendix A13): same state_dict keys."""

    def __init__(self, embed_dim=512, seq_len=6, output_layers=[256, 128, 64], nhead=8, num_layers=8,
                 ff_dim_factor=4):
        super().__init__()
        self.positional_encoding = PositionalEncoding(embed_dim, max_seq_len=seq_len)
        self.sa_layer = nn.TransformerEncoderLayer(d_model=embed_dim, nhead=nhead,
                                                   dim_feedforward=ff_dim_factor * embed_dim, activation="gelu",
                                                   batch_first=True, norm_first=True)
        self.sa_decoder = HipTransformerEncoder(self.sa_layer, num_layers=num_layers)
        # ReLU after EVERY layer incl. the last (tact.py:155-157), fused into each layer's epilogue
        self.output_layers = nn.ModuleList([HipLinear(seq_len * embed_dim, embed_dim, act='relu')])
        self.output_layers.append(HipLinear(embed_dim, output_layers[0], act='relu'))
        for i in range(len(output_layers) - 1):
            self.output_layers.append(HipLinear(output_layers[i], output_layers[i + 1], act='relu'))

    def forward(self, x, tail=(), pos_added=False):
        """``tail``: further HipLinear layers applied behind the output stack (the action head) -- one autograd node, one
        native backward call for the whole chain (hip_linear.mlp_chain).  ``pos_added``: the caller's token concatenation
        already added the positional encoding (``join_tokens``)."""
        if not pos_added:
            x = self.positional_encoding(x)
        x = self.sa_decoder(x)
        x = x.reshape(x.shape[0], -1)
        return run_chain(x, self.output_layers, tail)


def join_cols(parts, add=None):
    """``torch.cat(parts, dim=-1)`` of (B, w_i) tensors (+ ``add``, one row of the joined width) as one native launch
    forward and one backward (torch.ops.mi355ppo.cat_cols / split_cols); plain ``torch.cat`` off the device path's
    dtype (fp32) or beyond eight parts."""
    if len(parts) <= 8 and all(p.is_cuda and p.dtype is torch.float32 and p.dim() == 2 for p in parts):
        return torch.ops.mi355ppo.cat_cols([p.contiguous() for p in parts], add)
    out = torch.cat(parts, dim=-1)
    return out if add is None else out + add


class MLPDecoder(nn.Module):
    """tact.py:197-212"""

    def __init__(self, input_dim, hidden_layers, output_dim):
        super().__init__()
        layers, in_dim = [], input_dim
        for hidden_dim in hidden_layers:      # Identity keeps the ReLU's slot: same state_dict indices
            layers += [HipLinear(in_dim, hidden_dim, act='relu'), nn.Identity()]
            in_dim = hidden_dim
        layers.append(HipLinear(in_dim, output_dim))
        self.decoder = nn.Sequential(*layers)

    def forward(self, x, tail=()):
        return run_chain(x.reshape(x.shape[0], -1), self.decoder, tail)


class MultiModalModel(nn.Module):
    def __init__(self, context_size: int = 3, num_channels: int = 3, num_lin_features: int = 10,
                 num_outputs: int = 5, share_encoding: Optional[bool] = True, stack_tactile: Optional[bool] = True,
                 tactile_encoder: Optional[str] = "depth", img_encoder: Optional[str] = "depth",
                 seg_encoder: Optional[str] = "depth", tactile_encoding_size: Optional[int] = 128,
                 img_encoding_size: Optional[int] = 128, seg_encoding_size: Optional[int] = 128,
                 lin_encoding_size: Optional[int] = 128, mha_num_attention_heads: Optional[int] = 2,
                 mha_num_attention_layers: Optional[int] = 2, mha_ff_dim_factor: Optional[int] = 4,
                 include_lin: Optional[bool] = True, include_img: Optional[bool] = True,
                 include_seg: Optional[bool] = True, include_tactile: Optional[bool] = True,
                 include_pcl: Optional[bool] = False, additional_lin: Optional[int] = 0,
                 only_bc: Optional[bool] = False, pcl_conf: Optional[Dict] = None,
                 use_transformer: Optional[bool] = True) -> None:
        super().__init__()
        if (include_img and img_encoder != "depth") or (include_seg and seg_encoder != "depth"):
            raise NotImplementedError("efficientnet image encoders are outside the scope table (SURVEY section 2)")
        if tactile_encoder != "depth" or not stack_tactile or not share_encoding or additional_lin:
            raise NotImplementedError("only the reference's default 'depth' tactile encoder path is built")
        self.context_size = context_size
        self.num_output_params = num_outputs
        self.tactile_encoding_size = tactile_encoding_size
        self.alpha = 0.0
        self.num_lin_features = num_lin_features
        self.num_channels = 3
        self.stack_tactile = stack_tactile
        self.include_lin, self.include_tactile, self.include_pcl = include_lin, include_tactile, include_pcl
        self.include_img, self.include_seg = include_img, include_seg
        self.img_encoding_size, self.seg_encoding_size = img_encoding_size, seg_encoding_size
        self.pcl_conf = pcl_conf
        num_features = 0
        if include_tactile:
            self.tactile_encoder = CNNWithSpatialSoftArgmax(latent_dim=tactile_encoding_size)
            self.compress_tac_enc = nn.Identity()
            num_features += 1
        if include_img:                          # tact.py:299-315
            self.img_encoder = DepthOnlyFCBackbone54x96(latent_dim=img_encoding_size, num_channel=1)
            self.compress_img_enc = nn.Identity()
            num_features += 1
        if include_seg:                          # tact.py:317-333
            self.seg_encoder = DepthOnlyFCBackbone54x96(latent_dim=seg_encoding_size, num_channel=1)
            self.compress_seg_enc = nn.Identity()
            num_features += 1
        if include_lin:
            self.lin_encoding_size = lin_encoding_size
            self.lin_encoder = nn.Sequential(HipLinear(num_lin_features, 64, act='relu'), nn.Identity(),
                                             HipLinear(64, lin_encoding_size))
            num_features += 1
        if include_pcl:
            pcl_objects = 0
            self.pcl_encoder = nn.ModuleDict()
            for flag, name in (('merge_plug', 'plug_encoder'), ('merge_socket', 'socket_encoder'),
                               ('merge_goal', 'goal_encoder'), ('scene_pcl', 'scene_encoder')):
                if pcl_conf[flag]:
                    self.pcl_encoder[name] = PointNet()
                    pcl_objects += 1
            self.pcl_encoding_size = 256
            self.compress_pcl_enc = nn.Sequential(HipLinear(pcl_objects * self.pcl_encoding_size, 64, act='relu'),
                                                  nn.Identity(), HipLinear(64, lin_encoding_size))
            num_features += 1
        if use_transformer and (context_size > 1 or include_tactile):
            self.decoder = MultiLayerDecoder(embed_dim=tactile_encoding_size, seq_len=context_size * num_features,
                                             output_layers=[256, 128, 64, 32], nhead=mha_num_attention_heads,
                                             num_layers=mha_num_attention_layers, ff_dim_factor=mha_ff_dim_factor)
        else:
            self.decoder = MLPDecoder(input_dim=context_size * num_features * tactile_encoding_size,
                                      hidden_layers=[256, 128, 64], output_dim=32)
        self.latent_predictor = nn.Sequential(HipLinear(32, num_outputs, act='tanh' if only_bc else None),
                                              nn.Identity())
        self.reset_parameters()

    def reset_parameters(self):
        """tact.py:414-419: every nn.Linear re-initialised trunc_normal(0.02), zero bias."""
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.trunc_normal_(m.weight, std=.02)
                if m.bias is not None:
                    nn.init.zeros_(m.bias)

    def forward(self, obs_tactile=None, obs_img=None, obs_seg=None, lin_input=None, obs_pcl=None,
                add_lin_input=None):
        tokens_list = []
        if self.include_tactile:
            B, T, Fg, C, W, H = obs_tactile.shape                      # tact.py:429-432
            enc = self.tactile_encoder(obs_tactile.reshape(B * T, Fg * C, W, H))
            enc = enc.reshape((self.context_size, -1, self.tactile_encoding_size))
            tokens_list.append(torch.transpose(enc, 0, 1))
        if self.include_img:                     # tact.py:470-495
            B, T, C, W, H = obs_img.shape
            enc = self.img_encoder(obs_img.reshape(B * T, C, W, H))
            tokens_list.append(torch.transpose(enc.reshape((self.context_size, -1, self.img_encoding_size)), 0, 1))
        if self.include_seg:                     # tact.py:497-522
            B, T, C, W, H = obs_seg.shape
            enc = self.seg_encoder(obs_seg.reshape(B * T, C, W, H))
            tokens_list.append(torch.transpose(enc.reshape((self.context_size, -1, self.seg_encoding_size)), 0, 1))
        if self.include_lin:
            if lin_input.dim() == 2:
                lin_input = lin_input.reshape((lin_input.shape[0], self.context_size, self.num_lin_features))
            lin_encoding = run_chain(lin_input, self.lin_encoder)
            if lin_encoding.dim() == 2:
                lin_encoding = lin_encoding.unsqueeze(1)
            tokens_list.append(lin_encoding)
        if self.include_pcl:
            c = self.pcl_conf                                            # tact.py:542-566
            # the objects are consecutive slices of obs_pcl, in this order, each with its own PointNet
            objs = [(name, c[count]) for flag, name, count in (('merge_plug', 'plug_encoder', 'num_sample_plug'),
                                                               ('merge_socket', 'socket_encoder', 'num_sample_hole'),
                                                               ('merge_goal', 'goal_encoder', 'num_sample_goal'),
                                                               ('scene_pcl', 'scene_encoder', 'num_sample_all')) if c[flag]]
            if _PCL_ONE_LAUNCH and 2 <= len(objs) <= 4 and obs_pcl.is_cuda and obs_pcl.dtype is torch.float32:
                # every object in ONE forward (and one backward) launch, encodings written concatenated (round 6)
                enc, _idx = torch.ops.mi355ppo.pointnet_max_fwd_multi(
                    obs_pcl, [self.pcl_encoder[name].flat_parameters() for name, _n in objs], [int(n) for _name, n in objs])
            else:
                parts, NP = [], 0
                for name, n in objs:
                    parts.append(self.pcl_encoder[name](obs_pcl[:, NP:NP + n]))
                    NP += n
                enc = join_cols(parts)
            pcl_encoding = run_chain(enc, self.compress_pcl_enc)
            if pcl_encoding.dim() == 2:
                pcl_encoding = pcl_encoding.unsqueeze(1)
            tokens_list.append(pcl_encoding)
        # decoder output stack + action head as one chain (tact.py:155-157, 407-410)
        pos = getattr(self.decoder, 'positional_encoding', None)
        B, S, D = tokens_list[0].shape[0], sum(t.shape[1] for t in tokens_list), tokens_list[0].shape[2]
        if pos is not None and S <= pos.pos_enc.shape[1] and D == pos.pos_enc.shape[2] and \
                all(t.dim() == 3 and t.shape[0] == B and t.shape[2] == D for t in tokens_list):
            # torch.cat(tokens_list, dim=1) + the positional encoding in one launch (tokens are D-wide rows: joining the
            # (B, T_i * D) matrices column-wise is the concatenation along the token axis)
            tokens = join_cols([t.reshape(B, -1) for t in tokens_list], pos.pos_enc[0, :S].reshape(-1)).reshape(B, S, D)
            return self.decoder(tokens, tail=self.latent_predictor, pos_added=True)
        tokens = torch.cat(tokens_list, dim=1)
        return self.decoder(tokens, tail=self.latent_predictor)
