"""Pin the CPU oracle (oracle/teacher.py) against golden vectors captured from the reference's
own PPO.train_epoch (tests/golden/make_golden_teacher.py).  CPU only."""
import numpy as np
import pytest
import torch

from oracle import teacher as ot
from tests.golden_io import load_teacher, rollout


def _mk(meta, init, g):
    return ot.TeacherOracle(init, torch.from_numpy(g["perm"]), meta["num_envs"], meta["horizon"],
                            meta["mini_epochs"], meta["units"], meta["priv_units"])


@pytest.mark.parametrize("case", ["small", "default"])
def test_oracle_matches_reference(case):
    torch.set_num_threads(1)
    g, meta, init = load_teacher(case)
    # parameter layout is the reference's state_dict order
    shapes = ot.teacher_param_shapes(15, 64, 6, meta["units"], meta["priv_units"])
    assert list(shapes.keys()) == list(init.keys())
    assert all(tuple(init[k].shape) == s for k, s in shapes.items())
    orc = _mk(meta, init, g)
    for u in range(meta["n_updates"]):
        d = orc.prepare(rollout(g, u))
        np.testing.assert_allclose(orc.returns_raw.numpy(), g[f"u{u}/returns_raw"], rtol=0, atol=0)
        np.testing.assert_allclose(d["advantages"].numpy(), g[f"u{u}/advantages"], rtol=1e-6, atol=1e-6)
        np.testing.assert_allclose(d["values"].numpy(), g[f"u{u}/values_norm"], rtol=1e-6, atol=1e-6)
        np.testing.assert_allclose(d["returns"].numpy(), g[f"u{u}/returns_norm"], rtol=1e-6, atol=1e-6)
        vms = g[f"u{u}/vms_after_tail"]
        np.testing.assert_allclose([orc.rms_val.mean.item(), orc.rms_val.var.item(), orc.rms_val.count.item()],
                                   vms, rtol=1e-12)
        st = orc.update(record_grads=1)
        np.testing.assert_allclose(st["grads"][0].numpy(), g[f"u{u}/grad_step0"], rtol=1e-5, atol=1e-8)
        for name in ["a_losses", "c_losses", "b_losses", "entropies", "kls", "grad_total_norms",
                     "param_norms"]:
            got = np.array([x.item() for x in st[name]], dtype=np.float32)
            np.testing.assert_allclose(got, g[f"u{u}/{name}"], rtol=2e-5, atol=1e-7, err_msg=name)
        np.testing.assert_allclose(orc.flat_params().numpy(), g[f"u{u}/params_after"], rtol=0, atol=2e-6)
        np.testing.assert_allclose(orc.data["mus"].numpy(), g[f"u{u}/mus_after"], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(orc.data["sigmas"].numpy(), g[f"u{u}/sigmas_after"], rtol=1e-6)
        for nm, rs in [("running_mean_std", orc.rms_obs), ("priv_mean_std", orc.rms_priv),
                       ("value_mean_std", orc.rms_val)]:
            np.testing.assert_allclose(rs.mean.numpy(), g[f"u{u}/{nm}/running_mean"], rtol=1e-10, atol=1e-12)
            np.testing.assert_allclose(rs.var.numpy(), g[f"u{u}/{nm}/running_var"], rtol=1e-10)
            assert rs.count.item() == g[f"u{u}/{nm}/count"].item()
