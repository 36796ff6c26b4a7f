import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import tests.test_gpu_student as t
G = t.G
for tag in ["tac_pcl_lin", "tac_lin", "lin", "lin_latent"]:
    agent, env, (n, T, E) = t._agent(tag)
    model = agent.student.model
    teacher = {k[len(tag) + 9:]: torch.from_numpy(G[k]) for k in G.files if k.startswith(f"{tag}/teacher/")}
    if teacher: agent.agent.load_state_dict(teacher)
    stored = {k[len(tag) + 6:]: torch.from_numpy(G[k]) for k in G.files if k.startswith(f"{tag}/init/")}
    init = {k: (stored[k] if k in stored else t.big_weight(k, v.shape, t.SEEDS[tag])) for k, v in model.state_dict().items()}
    model.load_state_dict(init)
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout): m.p = 0.0
        if isinstance(m, torch.nn.MultiheadAttention): m.dropout = 0.0
    for k in agent.storage.storage_dict:
        agent.storage.storage_dict[k].copy_(torch.from_numpy(G[f"{tag}/in/{k}"]))
    agent.storage.indices.copy_(torch.from_numpy(G[f"{tag}/perm"]))
    agent.storage.prepare_training()
    agent.set_student_train()
    grad0 = {}
    def probe(step, m):
        if step == 0:
            grad0.update({k: p.grad.detach().clone() for k, p in m.named_parameters() if p.requires_grad and p.grad is not None})
    agent.grad_probe = probe
    agent.update()
    torch.cuda.synchronize()
    print("==", tag)
    for key in [k for k in G.files if k.startswith(f"{tag}/grad0/")]:
        nm = key[len(tag) + 7:]
        ref = G[key]; got = grad0[nm].cpu().numpy()
        e = np.abs(got - ref)
        print(f"{nm:60s} max|ref|={np.abs(ref).max():.3e} maxerr={e.max():.3e} rel={e.max()/max(np.abs(ref).max(),1e-30):.2e} relL2={np.linalg.norm(got-ref)/max(np.linalg.norm(ref),1e-30):.2e}")
