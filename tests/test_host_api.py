"""CPU-side checks: the C-ABI library loads and exports every symbol include/igi_ppo.h declares (no
compute without a GPU), the reference-shaped classes expose the reference's names / state_dict keys /
dtypes, and the product fails loudly (no CPU fallback) when asked to compute without a HIP device."""
import os
import re

import numpy as np
import pytest
import torch

from tests.golden_io import load_teacher

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from isaacgyminsertion_amd import _lib
    L = _lib.lib()
    hdr = open(os.path.join(ROOT, "include", "igi_ppo.h")).read()
    declared = set(re.findall(r"\b(igi_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"igi_stream_t"}
    assert declared, "no declarations parsed"
    for name in sorted(declared):
        assert hasattr(L, name), f"libigi_hip.so does not export {name}"
    assert declared == set(_lib.exported_symbols()), declared ^ set(_lib.exported_symbols())
    assert L.igi_abi_version() == _lib.ABI_VERSION


def test_param_layout_matches_state_dict_order():
    from isaacgyminsertion_amd.teacher_native import make_cfg, param_layout, teacher_param_shapes
    cfg, _ = make_cfg(15, 64, 6, [512, 256, 128], [256, 128, 8], 4096, 32, 8)
    total, layout = param_layout(cfg)
    shapes = teacher_param_shapes(15, 64, 6, [512, 256, 128], [256, 128, 8])
    assert len(layout) == len(shapes) == 23
    assert sum(s for _, s in layout) == 404501
    assert all(off % 4 == 0 for off, _ in layout)          # 16-byte aligned tensors
    assert all(int(np.prod(sh)) == sz for sh, (_, sz) in zip(shapes.values(), layout))
    offs = [o for o, _ in layout]
    assert offs == sorted(offs) and total >= offs[-1] + layout[-1][1]


def test_actor_critic_state_dict_is_the_reference_layout():
    from isaacgyminsertion_amd.algo.models.models_split import ActorCriticSplit
    g, meta, init = load_teacher("default")
    torch.manual_seed(42)
    m = ActorCriticSplit(dict(actor_units=meta["units"], actions_num=6, input_shape=(15,),
                              priv_mlp_units=meta["priv_units"], priv_info_dim=64, priv_info=True))
    sd = m.state_dict()
    assert list(sd.keys()) == list(init.keys())
    for k in sd:
        assert tuple(sd[k].shape) == tuple(init[k].shape) and sd[k].dtype == torch.float32
        # same recipe + same RNG stream as the reference (orthogonal init differs only by LAPACK threading)
        np.testing.assert_allclose(sd[k].numpy(), init[k].numpy(), atol=2e-6)
    # parameters are views of one flat vector; load_state_dict keeps it that way
    base = m.flat_params.data_ptr()
    m.load_state_dict(init)
    assert m.flat_params.data_ptr() == base
    for p in m.parameters():
        assert base <= p.data_ptr() < base + m.flat_params.numel() * 4
    np.testing.assert_array_equal(m.state_dict()["mu.weight"].numpy(), init["mu.weight"].numpy())
    with pytest.raises(RuntimeError):
        m.act({"obs": torch.zeros(2, 15), "priv_info": torch.zeros(2, 64)})   # no CPU fallback


def test_running_mean_std_state_dict_and_cpu_refusal():
    from isaacgyminsertion_amd.algo.models.running_mean_std import RunningMeanStd
    r = RunningMeanStd((15,))
    sd = r.state_dict()
    assert list(sd.keys()) == ["running_mean", "running_var", "count"]
    assert sd["running_mean"].dtype == torch.float64 and sd["count"].shape == ()
    assert sd["running_var"].sum().item() == 15 and sd["count"].item() == 1.0
    r.load_state_dict({"running_mean": torch.arange(15.).double(), "running_var": 2 * torch.ones(15).double(),
                       "count": torch.tensor(7.).double()})
    assert r.packed[14].item() == 14 and r.packed[15].item() == 2 and r.packed[30].item() == 7
    with pytest.raises(RuntimeError):
        r(torch.zeros(4, 15))


def test_config_access_semantics_and_trainer_refuses_cpu():
    from isaacgyminsertion_amd.utils.config import default_config
    from isaacgyminsertion_amd.algo.ppo.frozen_ppo import PPO, AdaptiveScheduler
    cfg = default_config(num_envs=64, horizon_length=8, rl_device="cpu")
    assert cfg.train.ppo.multi_gpu is False and cfg["rl_device"] == "cpu" and cfg.train.ppo.get("weight_decay", 0.0) == 0.0
    assert cfg.train.ppo["e_clip"] == 0.2 and cfg.task.env.numStates == 64
    with pytest.raises(RuntimeError):
        PPO(None, None, cfg)
    s = AdaptiveScheduler(0.02)
    assert s.update(1e-3, 0.1) == pytest.approx(1e-3 / 1.5) and s.update(1e-3, 0.001) == pytest.approx(1.5e-3)


def test_average_scalar_meter():
    from isaacgyminsertion_amd.utils.misc import AverageScalarMeter
    m = AverageScalarMeter(4)
    m.update(torch.tensor([1.0, 3.0]))
    m.update(torch.tensor([5.0, 5.0, 5.0]))
    assert len(m) == 4 and m.get_mean() == pytest.approx((2.0 * 1 + 5.0 * 3) / 4)


# Register ceilings of every kernel on a bench path (arch VGPRs + AGPRs per lane: one 512-entry file per SIMD, waves per SIMD
# = 512 / allocation) -- the ceiling is what the kernel's residency plan needs, not what it happens to use today -- and the
# scratch each may use (0 unless a row says otherwise, with the reason).  First matching row wins.
_RESOURCE_TABLE = [
    # (regex on the demangled name, max VGPR + AGPR, max scratch bytes / lane, why)
    (r"gemm_dma_kernel<256,", 256, 0, "the 256-wide 3-stage variants: one workgroup per CU by design"),
    (r"gemm_dma", 128, 0, "two workgroups of 8 waves per CU"),
    (r"k_trunk_loss", 128, 0, "the GEMM body + the loss epilogue: two workgroups per CU"),
    (r"k_rb_level<", 256, 0, "persistent row-block levels: one 512-thread workgroup per CU"),
    (r"k_env_fwd", 128, 0, "one wave per SIMD, 148 KB of LDS"),
    (r"k_policy_fwd", 256, 0, "persistent rollout policy kernel: 512 threads, one workgroup per CU"),
    (r"k_fwd12", 168, 0, "env_mlp + first trunk layer, 768 threads: three waves per SIMD"),
    (r"k_mlp_fwd", 256, 0, "eight waves per workgroup"),
    (r"k_token_fwd<", 128, 0, "1024-thread workgroups: four waves per SIMD"),
    (r"k_token_bwd<[1-4]>", 256, 0, "512-thread workgroups"),
    (r"k_token_bwd<", -1, 0, "S >= 5 must not be instantiated (88 - 940 bytes of scratch per lane in round 5): token_backward gates them"),
    (r"k_pointnet_fwd", 256, 0, "two 256-thread workgroups per CU"),
    (r"k_pointnet_bwd", 128, 0, "1024-thread workgroups"),
    (r"k_softargmax_|k_ssa_", 64, 0, "bandwidth kernels: eight waves per SIMD"),
    (r"k_slab_reduce_norm", 72, 0, "the norm-fusion variant (off by default): seven waves per SIMD"),
    (r"k_slab_reduce|k_sumsq_stats|k_clip_adam|k_adam_gather|k_gather_normalize|k_gather_rows|k_cat_cols", 64, 0,
     "bandwidth kernels: eight waves per SIMD"),
    (r"k_latent_bwd<\d, true>", 128, 0, "the row-dot path of the teacher step"),
    (r"k_loss", 256, 0, "fallback loss kernels"),
    (r"k_depth_", 128, 0, "depth backbone"),
    (r".", 512, 0, "every other kernel of the library: no scratch"),
]


def _kernel_resources():
    """{demangled kernel name: (vgpr, agpr, scratch)} of the whole library, from hipcc's kernel-resource-usage remarks"""
    import re
    import shutil
    import subprocess
    import tempfile
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = os.path.join(root, "isaacgyminsertion_amd", "csrc", "igi_capi.hip")
    with tempfile.TemporaryDirectory() as d:
        r = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-std=c++17", "-c",
                            "-Rpass-analysis=kernel-resource-usage", "-o", os.path.join(d, "x.o"), src],
                           capture_output=True, text=True, cwd=d)
    assert r.returncode == 0, r.stderr[-2000:]
    blocks = re.split(r"remark: [^\n]*Function Name: ", r.stderr)[1:]
    names = [b.split()[0] for b in blocks]
    filt = shutil.which("c++filt")
    dem = subprocess.run([filt], input="\n".join(names), capture_output=True, text=True).stdout.split("\n") if filt else names
    out = {}
    for b, mangled, d in zip(blocks, names, dem):
        vgpr = int(re.search(r"VGPRs: (\d+)", b).group(1))
        m = re.search(r"AGPRs: (\d+)", b)
        scratch = int(re.search(r"ScratchSize \[bytes/lane\]: (\d+)", b).group(1))
        name = re.sub(r"\(.*", "", d.replace("igi::", "").replace("void ", "")) if filt else mangled
        out[name] = (vgpr, int(m.group(1)) if m else 0, scratch)
    return out


def test_kernels_on_the_bench_paths_do_not_spill():
    """Resource-usage guard over EVERY kernel a bench path launches (round 5 looked at the GEMM kernels only and missed
    k_token_bwd<5..8> at 88 - 940 bytes of scratch per lane): no scratch, and a register ceiling per kernel from the table
    above.  A spill in the grouped weight-gradient kernel once cost 10 % of the headline metric without failing any
    numerical test."""
    import re
    res = _kernel_resources()
    seen = {pat: 0 for pat, *_ in _RESOURCE_TABLE}
    for name, (vgpr, agpr, scratch) in res.items():
        for pat, max_regs, max_scratch, why in _RESOURCE_TABLE:
            if re.search(pat, name):
                seen[pat] += 1
                assert max_regs >= 0, f"{name} must not be built: {why}"
                assert scratch <= max_scratch, (name, scratch, why)
                assert vgpr + agpr <= max_regs, (name, vgpr, agpr, max_regs, why)
                break
    assert seen[r"gemm_dma"] >= 20 and seen[r"k_rb_level<"] >= 3 and seen[r"k_token_bwd<[1-4]>"] == 4, seen
    for pat in (r"k_trunk_loss", r"k_env_fwd", r"k_mlp_fwd", r"k_token_fwd<", r"k_pointnet_fwd", r"k_pointnet_bwd"):
        assert seen[pat] >= 1, (pat, seen)


def test_pcl_augmentations():
    """factory_utils.py:83-165 semantics on CPU tensors (pure elementwise torch, device-agnostic)."""
    from isaacgyminsertion_amd.envs.pcl_augment import PointCloudAugmentations
    aug = PointCloudAugmentations()
    torch.manual_seed(0)
    pts = torch.randn(5, 400, 3) * 0.05
    out = aug.augment(pts.clone(), None, None, torch.full((5, 1, 3), 0.5))
    d = out - pts
    assert d.abs().max() <= 2 * aug.noise_clip + 1e-9                     # jitter and offset are each clipped
    assert torch.allclose(d.mean(dim=1), torch.full((5, 3), 0.0005), atol=1e-4)   # const_noise * pcl_noise
    moved = ((d - 0.0005).abs() > 1e-9).any(-1).float().mean().item()
    assert 0.2 < moved < 0.4                                               # noise_prob = 0.3
    # rotation: about z by 90 degrees maps (1,0,0) -> (0,-1,0) for ROW vectors p @ R
    p = torch.tensor([[[1.0, 0.0, 0.0]], [[1.0, 0.0, 0.0]]])
    r = aug.random_rotate(p, torch.tensor([torch.pi / 2, torch.pi / 2]), torch.tensor([2, 0]))
    assert torch.allclose(r[0], torch.tensor([[0.0, -1.0, 0.0]]), atol=1e-6)
    assert torch.allclose(r[1], torch.tensor([[1.0, 0.0, 0.0]]), atol=1e-6)
    s = aug.random_scale_anisotropic(torch.ones(4, 10, 3))
    assert (s >= 0.8).all() and (s <= 1.2).all() and torch.equal(s[:, 0], s[:, 9])
    dr = aug.batch_random_dropout(torch.ones(2000, 50, 3))
    per_cloud = (dr == 0).all(-1).float().mean(1)
    assert 0.7 < (per_cloud == 0).float().mean().item() < 0.9             # 80 % of the clouds are exempt
    o = aug.add_outliers(torch.zeros(3, 400, 3))
    assert 20 <= (o != 0).any(-1).sum(1).min().item() <= 40


def test_student_ops_refuse_cpu_tensors():
    """No CPU / eager fallback anywhere on the student path either: every native op raises on host tensors."""
    from isaacgyminsertion_amd.algo.models.transformer.depth_backbone import DepthOnlyFCBackbone54x96
    from isaacgyminsertion_amd.algo.models.transformer.pointnets import PointNet
    from isaacgyminsertion_amd.algo.models.transformer.tactile_cnn import CNNWithSpatialSoftArgmax
    from isaacgyminsertion_amd.hip_linear import HipLinear
    from isaacgyminsertion_amd.hip_token_encoder import HipTransformerEncoder
    layer = torch.nn.TransformerEncoderLayer(d_model=32, nhead=2, dim_feedforward=128, activation="gelu",
                                             batch_first=True, norm_first=True)
    cases = [
        (HipLinear(15, 64, act="relu"), torch.zeros(4, 15)),
        (HipTransformerEncoder(layer, 2), torch.zeros(4, 3, 32)),
        (DepthOnlyFCBackbone54x96(latent_dim=32), torch.zeros(2, 1, 54, 96)),
        (PointNet(), torch.zeros(2, 400, 3)),
        (CNNWithSpatialSoftArgmax(latent_dim=32), torch.zeros(2, 3, 32, 64)),
    ]
    for module, x in cases:
        with pytest.raises(RuntimeError, match="HIP device only"):
            module(x)


def test_deploy_surface_and_replay_robot():
    """The deployment players keep the reference's module / class names (algo/deploy/deploy_s1.py, deploy_s2.py);
    the robot side enters through RobotIO.  No compute here (no GPU)."""
    from isaacgyminsertion_amd.algo.deploy import RobotIO, ReplayRobot
    from isaacgyminsertion_amd.algo.deploy import deploy_s1, deploy_s2
    from isaacgyminsertion_amd.utils.config import default_config
    for mod, names in ((deploy_s1, ("restore", "set_eval", "policy_step", "deploy")),
                       (deploy_s2, ("restore", "restore_student", "set_eval", "set_student_eval", "process_obs",
                                    "policy_step", "deploy"))):
        for n in names:
            assert callable(getattr(mod.HardwarePlayer, n))
    cfg = default_config()
    assert cfg.deploy.ppo.tactile_info is True and cfg.deploy.rl.max_episode_length == 500
    frames = [{"obs": torch.full((1, 2), float(i))} for i in range(3)]
    r = ReplayRobot(frames, episode_length=2)
    assert isinstance(r, RobotIO) and not r.done()
    assert r.observe()["obs"][0, 0] == 0
    r.apply(torch.zeros(1, 6))
    assert r.observe()["obs"][0, 0] == 1 and not r.done()
    r.apply(torch.ones(1, 6))
    assert r.done() and r.stacked_actions().shape == (2, 1, 6)
    with pytest.raises(NotImplementedError):
        RobotIO().observe()


def test_header_is_plain_c(tmp_path):
    """include/igi_ppo.h is the drop-in boundary: it has to compile as strict C99 (no C++ / torch types), so any FFI
    (ctypes, cgo, JNI, ...) can bind it."""
    import shutil
    import subprocess
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("no gcc")
    hdr = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "igi_ppo.h")
    src = tmp_path / "hdr.c"
    src.write_text(f'#include "{hdr}"\nint main(void) {{ return (int)sizeof(igi_teacher_cfg) == 0; }}\n')
    r = subprocess.run([gcc, "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", str(src)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_average_scalar_meter_device_sums_equal_batches():
    """update_sums (per-step device sums from the rollout kernel) folds to the same windowed mean as update()."""
    from isaacgyminsertion_amd.utils.misc import AverageScalarMeter
    a, b = AverageScalarMeter(10), AverageScalarMeter(10)
    g = torch.Generator().manual_seed(0)
    sums, counts = [], []
    for _ in range(40):
        k = int(torch.randint(0, 4, (1,), generator=g))
        vals = torch.randn(k, 1, generator=g)
        a.update(vals)
        sums.append(vals.sum().reshape(1))
        counts.append(torch.tensor([float(k)]))
    b.update_sums(torch.cat(sums), torch.cat(counts))
    assert len(a) == len(b) == 10
    assert a.get_mean() == pytest.approx(b.get_mean(), rel=1e-5, abs=1e-6)


def test_reference_import_paths_resolve_to_this_package():
    """isaacgyminsertion/train.py:31-32, train_supervised.py:40 and the deploy scripts import ``algo.*``; with the repo
    root on PYTHONPATH those statements (verbatim) give this package's classes -- no source edit in train.py."""
    import subprocess
    import sys
    code = (
        "from algo.ppo.frozen_ppo import PPO\n"
        "from algo.ext_adapt.ext_adapt import ExtrinsicAdapt\n"
        "from algo.models.transformer.runner import Runner\n"
        "from algo.models.models_split import ActorCriticSplit as ActorCritic\n"
        "from algo.models.running_mean_std import RunningMeanStd\n"
        "from algo.ppo.experience import ExperienceBuffer, StudentBuffer\n"
        "from algo.models.transformer.tactile_cnn import CNNWithSpatialSoftArgmax\n"
        "from algo.models.transformer.pointnets import PointNet\n"
        "from algo.models.transformer.tact import MultiModalModel\n"
        "from algo.deploy.deploy_s1 import HardwarePlayer\n"
        "import algo, isaacgyminsertion_amd.algo.ppo.frozen_ppo as real\n"
        "assert PPO is real.PPO and PPO.__module__ == 'isaacgyminsertion_amd.algo.ppo.frozen_ppo'\n"
        "import isaacgyminsertion_amd.algo.models.transformer.runner as rr\n"
        "assert Runner is rr.Runner\n"
        "print('ok')\n")
    env = dict(os.environ, PYTHONPATH=ROOT)
    out = subprocess.run([sys.executable, "-c", code], cwd="/tmp", env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), out.stderr[-2000:]


def test_student_buffer_minibatches_are_the_reference_gather():
    """StudentBuffer.__getitem__ (experience.py:117-139: every key of the env-major flattened arena indexed with the fixed
    permutation's slice) through the lazy per-key gather from the time-major arena: same rows for every key, the dict
    interface the trainers use, and the cached arena rows follow an edit of ``indices``."""
    from isaacgyminsertion_amd.algo.ppo.experience import StudentBuffer
    torch.manual_seed(0)
    N, T, mb = 6, 5, 10
    buf = StudentBuffer(N, T, N * T, mb, obs_dim=4, act_dim=3, priv_dim=7,
                        student_dims={"tactile": (3, 8), "student_obs": 4, "pcl": (12,)}, device="cpu")
    for k, v in buf.storage_dict.items():
        v.copy_(torch.randn(v.shape))
    buf.prepare_training()
    assert len(buf) == 3

    def reference(i):
        b = buf.indices[i * mb:(i + 1) * mb]
        return {k: v.transpose(0, 1).flatten(0, 1)[b] for k, v in buf.storage_dict.items()}   # experience.py:141-145, :117-139

    for i in range(len(buf)):
        got, ref = buf[i], reference(i)
        assert set(got.keys()) == set(ref) and len(got) == len(ref) and "n_tactile" in got and "n_img" not in got
        assert got.get("n_img") is None
        for k, v in got.items():
            assert torch.equal(v, ref[k]), k
        assert torch.equal(got["n_tactile"], ref["n_tactile"]) and got["n_tactile"].shape == (mb, 3, 8)
        # env-major view of the whole arena (what ExtrinsicAdapt reads outside the minibatch loop)
        assert torch.equal(buf.data_dict["teacher_actions"], buf.storage_dict["teacher_actions"].transpose(0, 1).flatten(0, 1))
    buf.indices.copy_(buf.indices.flip(0))            # an in-place edit of the permutation is noticed
    assert torch.equal(buf[0]["n_obs"], reference(0)["n_obs"])
    buf.indices = torch.arange(N * T)                  # ... and so is a replaced tensor
    assert torch.equal(buf[1]["n_pcl"], reference(1)["n_pcl"])


def test_bench_parent_starts_ranks_as_a_child_and_returns_their_exit_code():
    """`python bench.py --gpus 2` with no WORLD_SIZE: the parent must start torch.distributed.run as a CHILD process and
    pass its exit code on -- here (no GPU in this container) the ranks fail, so the parent must exit non-zero, print no
    JSON line, and must not have initialised torch itself (it says which command it started)."""
    import subprocess
    import sys
    env = dict(os.environ, IGI_DIST_BACKEND="gloo", IGI_PG_TIMEOUT_S="60")
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                          "--no-cpu-baseline", "--no-roofline", "--no-multi-configs"], cwd=ROOT, env=env,
                         capture_output=True, text=True, timeout=600)
    if __import__("torch").cuda.is_available():
        pytest.skip("a GPU is present: the ranks would succeed (covered by tests/test_gpu_dp.py)")
    assert out.returncode != 0
    assert "starting 2 ranks" in out.stderr and "torch.distributed.run" in out.stderr
    assert not [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
