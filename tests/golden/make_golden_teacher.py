"""Generate teacher-PPO golden vectors from the REFERENCE implementation.

Runs ONLY in the build container (needs /root/reference).  It imports the
reference's own ``algo.ppo.frozen_ppo.PPO`` on CPU (see ``ref_harness``), fills the
reference ``ExperienceBuffer`` with a seeded synthetic rollout through the
reference's own ``model_act`` (exactly what ``play_steps`` stores,
frozen_ppo.py:655-683), then runs the reference's *unmodified*
``PPO.train_epoch`` (frozen_ppo.py:495-646), whose ``play_steps`` is replaced by
that fill + the reference's own tail (``computer_return``, ``prepare_training``,
value normalisation; frozen_ppo.py:714-725).

Outputs ``teacher_<case>.npz`` with inputs (rollout arena, permutation, initial
parameters) and expected outputs (GAE returns, normalised advantages/values,
per-step losses, first-step raw gradient, per-epoch KL, post-update parameters,
normaliser states, written-back mu/sigma) for one or two consecutive updates.

    python tests/golden/make_golden_teacher.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_harness as rh  # noqa: E402

rh.install()
from algo.ppo.frozen_ppo import PPO  # noqa: E402  (reference)


def synth_fill(agent, gen, done_p):
    """What play_steps stores (frozen_ppo.py:655-683) for a synthetic env."""
    st = agent.storage
    T, N = st.transitions_per_env, st.num_envs
    obs_dim, priv_dim = st.obs_dim, st.priv_dim
    for n in range(T):
        obs = {"obs": torch.randn(N, obs_dim, generator=gen),
               "priv_info": torch.randn(N, priv_dim, generator=gen)}
        res = agent.model_act(obs)
        st.update_data("obses", n, obs["obs"])
        st.update_data("priv_info", n, obs["priv_info"])
        for k in ["actions", "neglogpacs", "values", "mus", "sigmas"]:
            st.update_data(k, n, res[k])
        dones = (torch.rand(N, generator=gen) < done_p).to(torch.uint8)
        st.update_data("dones", n, dones)
        st.update_data("rewards", n, 0.1 * torch.randn(N, 1, generator=gen))
    obs = {"obs": torch.randn(N, obs_dim, generator=gen),
           "priv_info": torch.randn(N, priv_dim, generator=gen)}
    return agent.model_act(obs)["values"]


def ref_tail(agent, last_values):
    """The reference's own post-rollout tail (frozen_ppo.py:714-725), verbatim calls."""
    agent.storage.computer_return(last_values, agent.gamma, agent.tau)
    agent.storage.prepare_training()
    returns = agent.storage.data_dict["returns"]
    values = agent.storage.data_dict["values"]
    if agent.normalize_value:
        agent.value_mean_std.train()
        values = agent.value_mean_std(values)
        returns = agent.value_mean_std(returns)
        agent.value_mean_std.eval()
    agent.storage.data_dict["values"] = values
    agent.storage.data_dict["returns"] = returns


def flat_params(model):
    return torch.cat([p.detach().reshape(-1) for p in model.parameters()]).numpy().copy()


def run_case(name, num_envs, horizon, mini_epochs, units, priv_units, n_updates, done_p, seed=42,
             data_seed=1234):
    cfg = rh.teacher_config(num_envs, horizon, mini_epochs, units=units, priv_units=priv_units)
    torch.manual_seed(seed)
    agent = PPO(None, None, cfg)
    gen = torch.Generator().manual_seed(data_seed)
    out = {}
    out["meta"] = np.array([num_envs, horizon, mini_epochs, n_updates], dtype=np.int64)
    out["units"] = np.array(units, dtype=np.int64)
    out["priv_units"] = np.array(priv_units, dtype=np.int64)
    for k, v in agent.model.state_dict().items():
        out[f"init/{k}"] = v.numpy().copy()
    out["perm"] = agent.storage.indices.numpy().copy()

    for u in range(n_updates):
        rec = {"grads": [], "norms": []}

        def fake_play_steps(u=u):
            last_values = synth_fill(agent, gen, done_p)
            for k in ["obses", "priv_info", "rewards", "values", "neglogpacs", "dones", "actions",
                      "mus", "sigmas"]:
                out[f"u{u}/in/{k}"] = agent.storage.storage_dict[k].numpy().copy()
            out[f"u{u}/in/last_values"] = last_values.numpy().copy()
            ref_tail(agent, last_values)
            dd = agent.storage.data_dict
            out[f"u{u}/returns_raw"] = agent.storage.storage_dict["returns"].numpy().copy()  # (T,N,1)
            out[f"u{u}/advantages"] = dd["advantages"].numpy().copy()                          # (B,)
            out[f"u{u}/values_norm"] = dd["values"].numpy().copy()                              # (B,1)
            out[f"u{u}/returns_norm"] = dd["returns"].numpy().copy()                            # (B,1)
            out[f"u{u}/vms_after_tail"] = np.array(
                [agent.value_mean_std.running_mean.item(), agent.value_mean_std.running_var.item(),
                 agent.value_mean_std.count.item()], dtype=np.float64)

        agent.play_steps = fake_play_steps

        orig_clip = torch.nn.utils.clip_grad_norm_

        def rec_clip(params, max_norm, *a, **k):
            params = list(params)
            if len(rec["grads"]) < 2:  # raw (pre-clip) gradient of the first two optimizer steps
                rec["grads"].append(torch.cat([p.grad.reshape(-1) for p in params]).numpy().copy())
            n = orig_clip(params, max_norm, *a, **k)
            rec["norms"].append(float(n))
            return n

        torch.nn.utils.clip_grad_norm_ = rec_clip
        try:
            a_losses, c_losses, b_losses, entropies, kls, grad_norms, _ = agent.train_epoch()
        finally:
            torch.nn.utils.clip_grad_norm_ = orig_clip

        out[f"u{u}/a_losses"] = np.array([x.item() for x in a_losses], dtype=np.float32)
        out[f"u{u}/c_losses"] = np.array([x.item() for x in c_losses], dtype=np.float32)
        out[f"u{u}/b_losses"] = np.array([x.item() for x in b_losses], dtype=np.float32)
        out[f"u{u}/entropies"] = np.array([x.item() for x in entropies], dtype=np.float32)
        out[f"u{u}/kls"] = np.array([x.item() for x in kls], dtype=np.float32)
        out[f"u{u}/param_norms"] = np.array([x.item() for x in grad_norms], dtype=np.float32)
        out[f"u{u}/grad_total_norms"] = np.array(rec["norms"], dtype=np.float32)
        out[f"u{u}/grad_step0"] = rec["grads"][0]
        if name == "small":
            out[f"u{u}/grad_step1"] = rec["grads"][1]
        out[f"u{u}/params_after"] = flat_params(agent.model)
        out[f"u{u}/mus_after"] = agent.storage.data_dict["mus"].numpy().copy()
        out[f"u{u}/sigmas_after"] = agent.storage.data_dict["sigmas"].numpy().copy()
        for nm in ["running_mean_std", "priv_mean_std", "value_mean_std"]:
            m = getattr(agent, nm)
            out[f"u{u}/{nm}/running_mean"] = m.running_mean.numpy().copy()
            out[f"u{u}/{nm}/running_var"] = m.running_var.numpy().copy()
            out[f"u{u}/{nm}/count"] = np.array(m.count.item(), dtype=np.float64)

    path = os.path.join(HERE, f"teacher_{name}.npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path}: {os.path.getsize(path) / 1e6:.2f} MB, "
          f"a_loss[0]={out['u0/a_losses'][0]:.6f} c_loss[0]={out['u0/c_losses'][0]:.6f}")


if __name__ == "__main__":
    torch.set_num_threads(1)  # fixed reduction order for reproducible goldens
    # small network, two consecutive updates (state carry-over: Adam moments, normalisers)
    run_case("small", num_envs=32, horizon=8, mini_epochs=4, units=(64, 48, 32), priv_units=(48, 32, 8),
             n_updates=2, done_p=0.05)
    # reference default network dims (404,501 params), one update
    run_case("default", num_envs=64, horizon=8, mini_epochs=4, units=(512, 256, 128),
             priv_units=(256, 128, 8), n_updates=1, done_p=0.05)
