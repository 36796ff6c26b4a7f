"""ORACLE (test infrastructure, NOT product code): CPU restatement, in numpy fp32, of the reference's
offline supervised student step for the proprio-only model of BASELINE configs[0]
(``train_supervised.py`` -> ``Runner.train`` / ``Runner.validate``).

Only ``tests/`` and ``__graft_entry__.smoke()`` may import this package; the product path never does.

Parity status: PINNED against ``tests/golden/offline.npz`` (case ``cfg1``), captured from the reference's
own ``Runner.train`` on CPU by ``tests/golden/make_golden_offline.py``; see tests/test_oracle_offline.py.

What is restated (paths relative to /root/reference):
  * model  : ``lin_encoder`` Linear(15,64)-ReLU-Linear(64,32) (algo/models/transformer/tact.py:337-339,
             531-540), ``MLPDecoder`` 32->256->128->64->32 with ReLU between (tact.py:197-212),
             head Linear(32,6)+Tanh under only_bc (tact.py:407-410)
  * loss   : MSELoss(mean)(out, action[:, -1, :]) (runner.py:226-230, 580); validation clamps out to [-1, 1]
             first (runner.py:326)
  * update : zero_grad, backward, clip_grad_norm_(0.5) (eps 1e-6), AdamW(lr, weight_decay=1e-6) single-tensor
             rule with Python-double scalars (runner.py:243-248, 481)
"""
import math

import numpy as np

F32 = np.float32
LAYERS = [("lin_encoder.0", "relu"), ("lin_encoder.2", None), ("decoder.decoder.0", "relu"),
          ("decoder.decoder.2", "relu"), ("decoder.decoder.4", "relu"), ("decoder.decoder.6", None),
          ("latent_predictor.0", "tanh")]


def forward(params, x):
    """x (B, 15) -> out (B, 6) and the per-layer (input, output) pairs for backward."""
    tape, h = [], x.astype(F32)
    for name, act in LAYERS:
        z = h @ params[name + ".weight"].T + params[name + ".bias"]
        y = np.maximum(z, 0) if act == "relu" else (np.tanh(z) if act == "tanh" else z)
        tape.append((h, y.astype(F32)))
        h = y.astype(F32)
    return h, tape


def loss_and_grads(params, x, target):
    out, tape = forward(params, x)
    diff = out - target.astype(F32)
    loss = F32(np.mean(diff * diff))
    d = (F32(2.0) * diff / F32(diff.size)).astype(F32)
    grads = {}
    for (name, act), (inp, y) in zip(reversed(LAYERS), reversed(tape)):
        if act == "relu":
            d = d * (y > 0)
        elif act == "tanh":
            d = d * (F32(1) - y * y)
        grads[name + ".weight"] = (d.T @ inp).astype(F32)
        grads[name + ".bias"] = d.sum(0).astype(F32)
        d = (d @ params[name + ".weight"]).astype(F32)
    return loss, grads


def validate_loss(params, x, target):
    out, _ = forward(params, x)
    diff = np.clip(out, -1, 1) - target.astype(F32)
    return F32(np.mean(diff * diff))


class AdamW:
    """torch.optim.AdamW single-tensor step preceded by clip_grad_norm_ (runner.py:246-248)."""

    def __init__(self, params, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-6, max_norm=0.5):
        self.p = params
        self.lr, self.b1, self.b2, self.eps, self.wd, self.max_norm = lr, betas[0], betas[1], eps, weight_decay, max_norm
        self.m = {k: np.zeros_like(v) for k, v in params.items()}
        self.v = {k: np.zeros_like(v) for k, v in params.items()}
        self.t = 0

    def step(self, grads):
        total = F32(math.sqrt(sum(float(np.sum(g.astype(np.float64) ** 2)) for g in grads.values())))
        coef = min(F32(self.max_norm) / (total + F32(1e-6)), F32(1.0))
        self.t += 1
        bc1, bc2 = 1 - self.b1 ** self.t, 1 - self.b2 ** self.t
        step_size, bc2_sqrt = self.lr / bc1, math.sqrt(bc2)
        for k, p in self.p.items():
            g = (grads[k] * F32(coef)).astype(F32)
            p *= F32(1 - self.lr * self.wd)
            self.m[k] += F32(1 - self.b1) * (g - self.m[k])
            self.v[k] = (self.v[k] * F32(self.b2) + F32(1 - self.b2) * g * g).astype(F32)
            denom = (np.sqrt(self.v[k]) / F32(bc2_sqrt) + F32(self.eps)).astype(F32)
            p += F32(-step_size) * (self.m[k] / denom)
        return total


def train_epochs(init, stud_obs, action, val_obs, val_action, epochs=3, batch=64, lr=1e-4):
    """The golden's schedule: per epoch validate, then one pass over the frames in order; -> (train losses,
    val losses [validate, end-of-train-pass] per epoch, final params)."""
    params = {k: v.astype(F32).copy() for k, v in init.items()}
    opt = AdamW(params, lr=lr)
    train, val = [], []
    for _ in range(epochs):
        val.append(float(validate_loss(params, val_obs[:, -1], val_action[:, -1])))
        for s in range(0, stud_obs.shape[0], batch):
            loss, grads = loss_and_grads(params, stud_obs[s:s + batch, -1], action[s:s + batch, -1])
            opt.step(grads)
            train.append(float(loss))
        val.append(float(validate_loss(params, val_obs[:, -1], val_action[:, -1])))
    return np.array(train), np.array(val), params
