"""oracle/student.py (the CPU restatement of the assembled student: MultiModalModel forward + the behaviour-cloning loss)
against tests/golden/student.npz -- vectors the reference's own, unmodified ExtrinsicAdapt.train_epoch produced
(tests/golden/make_golden_student.py): the step-0 loss and the RAW step-0 gradient of every parameter, for the
tactile + pcl + lin, tactile + lin and lin-only students.  Same library (PyTorch CPU) on both sides: 1e-5."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
G = np.load(os.path.join(ROOT, "tests", "golden", "student.npz"))


def _minibatch0(tag):
    N, T, E = [int(v) for v in G[f"{tag}/flags"][:3]]
    perm = torch.from_numpy(G[f"{tag}/perm"]).long()
    ids = perm[: N * T // E]
    t, n = ids % T, ids // T                      # sample id b = n*T + t lives at arena[t, n] (experience.py:39-46)

    def rows(key):
        k = f"{tag}/in/{key}"
        if k not in G.files:
            return None
        a = torch.from_numpy(G[k])
        return a[t, n].reshape(len(ids), -1)
    return rows("n_tactile"), rows("n_student_obs"), rows("n_pcl"), rows("teacher_actions")


@pytest.mark.parametrize("tag", ["tac_pcl_lin", "tac_lin", "tac_lin_illcond", "lin"])
def test_student_oracle_reproduces_the_reference_step0(tag):
    from oracle import student as os_
    sd = {k[len(tag) + 6:]: torch.from_numpy(G[k]) for k in G.files if k.startswith(f"{tag}/init/")}
    tactile, obs, pcl, act = _minibatch0(tag)
    if tactile is not None:
        tactile = tactile.reshape(tactile.shape[0], 3, -1)
    loss, grads = os_.loss_and_grads(sd, act, tactile, obs, pcl)
    np.testing.assert_allclose(loss, float(G[f"{tag}/action_losses"][0]), rtol=1e-5)
    names = [k[len(tag) + 7:] for k in G.files if k.startswith(f"{tag}/grad0/")]
    assert len(names) >= 10
    gmax = max(np.abs(G[f"{tag}/grad0/{n}"]).max() for n in names)
    for n in names:
        ref = G[f"{tag}/grad0/{n}"]
        noise = float(G[f"{tag}/grad0_ref_noise/{n}"])
        np.testing.assert_allclose(grads[n].numpy(), ref, atol=max(1e-5 * np.abs(ref).max(), 1e-7 * gmax, 0.1 * noise),
                                   rtol=1e-4, err_msg=n)
    # parameters without a gradient in the reference (the never-used decoder.sa_layer.* template) have none here
    for n, g in grads.items():
        if n not in names:
            assert g is None or float(g.abs().max()) == 0.0, n
    m = os_.discontinuity_margin(sd, act, tactile, obs, pcl)
    assert m.shape == (act.shape[0],) and bool((m >= 0).all())
