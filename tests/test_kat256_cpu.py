"""Known answers at the production shape (hidden 256, batch 256, 4096 acting rows), produced by the imported reference
(tests/golden/gen_model_golden_256.py): the module / autograd path of this stack on the CPU.  The fused kernels the bench
times take the same fixture in tests/test_kat256_gpu.py."""
import numpy as np
import torch

import kat256 as KAT


def test_fixture_is_small_and_numeric():
    G = KAT.golden()
    assert len(G.files) > 300 and all(G[k].dtype.kind in "fiu" for k in G.files)
    assert G["sac.returns"].shape == (5,) and G["act.task_action"].shape == (KAT.K.N_ACT, 2)
    # the update moved something, and the gate at eps_safe splits the acting batch
    assert 0.05 < float((G["act.risk"] > 0.3).mean()) <= 1.0


def test_one_sac_and_one_qrisk_update_at_hidden_256_batch_256_match_the_reference():
    G = KAT.golden()
    agent = KAT.build_agent("cpu")
    qr = agent.safety_critic
    batch, c, e1, e2 = KAT.inputs("cpu")
    res = agent.update_parameters(None, KAT.K.B, 0, safety_critic=qr, batch=batch, eps_next=e1, eps_pi=e2, as_floats=True)
    assert np.allclose(res, G["sac.returns"], rtol=KAT.REL, atol=2e-6), (res, G["sac.returns"])
    KAT.check_grads(G, "sac.grad.critic", {k: p.grad for k, p in agent.critic.named_parameters()})
    KAT.check_grads(G, "sac.grad.policy", {k: p.grad for k, p in agent.policy.named_parameters()})
    KAT.check_post(G, "sac.post.critic", agent.critic, "sac.grad.critic")
    KAT.check_post(G, "sac.post.critic_target", agent.critic_target)
    KAT.check_post(G, "sac.post.policy", agent.policy, "sac.grad.policy")
    qbatch = (batch[0], batch[1], c, batch[3], batch[4])
    qr.update_parameters(policy=agent.policy, batch=qbatch, eps_next=e1, eps_pi=e2)
    KAT.check_grads(G, "mf.grad.qrisk", {k: p.grad for k, p in qr.safety_critic.named_parameters() if p.grad is not None})
    KAT.check_grads(G, "mf.grad.recpolicy", {k: p.grad for k, p in qr.policy.named_parameters()})
    KAT.check_post(G, "mf.post.qrisk", qr.safety_critic, "mf.grad.qrisk")
    KAT.check_post(G, "mf.post.qrisk_target", qr.safety_critic_target)
    KAT.check_post(G, "mf.post.recpolicy", qr.policy, "mf.grad.recpolicy")
    assert np.allclose(qr.get_value(batch[0], batch[1]).numpy().ravel(), G["mf.get_value"], rtol=KAT.REL, atol=2e-6)
    # acting pass on the updated networks (experiment.py:546-577)
    obs, noise = KAT.K.acting()
    obs, noise = torch.as_tensor(obs), torch.as_tensor(noise)
    with torch.no_grad():
        task, _, _ = agent.policy.sample(obs, noise[0])
        risk = qr.get_value(obs, task)
        rec, _, _ = qr.policy.sample(obs, noise[1])
    KAT.check_acting(G, task, risk, rec)
