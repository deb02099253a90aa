"""Generate the known-answer fixture for ONE optimiser step and for a 2-epoch run of the REFERENCE's
`MPC.train` (recovery_rl/MPC.py:213-309: bootstrap idxs :255-257, batch loop :266-298, loss :270-292) by IMPORTING
the reference in this container.

Run: python tests/golden/gen_mpc_train_golden.py -> tests/golden/mpc_train_golden.npz (data only).

The ensemble starts from the weights already stored in mpc_golden.npz (`pt.*`), so only the data, the bootstrap /
shuffle index tables the reference drew, the per-step losses, the gradients of the single step and the trained
parameters are stored.  The two 5x200x200 matrices are stored on a strided subset (every 29th element) to keep the
fixture small; everything else is stored whole.
"""
import contextlib
import io
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _ref_shims  # noqa: E402

_ref_shims.install()

import torch  # noqa: E402

BIG = ("lin1_w", "lin2_w")
STRIDE = 29
PARAMS = ("lin0_w", "lin0_b", "lin1_w", "lin1_b", "lin2_w", "lin2_b", "lin3_w", "lin3_b", "max_logvar", "min_logvar")


def subset(name, arr):
    arr = np.asarray(arr)
    return arr.reshape(-1)[::STRIDE].copy() if name in BIG else arr.copy()


def build_reference_mpc(G):
    from config import create_config
    from dotmap import DotMap
    from recovery_rl.MPC import MPC
    with contextlib.redirect_stdout(io.StringIO()):
        cfg = create_config("navigation2", "MPC", DotMap(), [], "/tmp")
        mpc = MPC(cfg.ctrl_cfg)
    sd = mpc.model.state_dict()
    mpc.model.load_state_dict({k: torch.as_tensor(G["pt." + k]) for k in sd})
    return mpc


def run_train(mpc, s, a, s2, epochs, idxs0):
    """Run the reference's train() with the bootstrap table injected; records the shuffled tables, the loss of
    every optimiser step and leaves the gradients of the LAST step in .grad."""
    import recovery_rl.MPC as rm
    rec = {"shuffled": [], "losses": []}
    real_shuffle, real_np, real_backward = rm.shuffle_rows, rm.np, torch.Tensor.backward

    class RandomProxy:
        def __getattr__(self, k):
            return getattr(np.random, k)

        def randint(self, n, size=None):
            assert list(size) == list(idxs0.shape) and n == idxs0.shape[1]
            return idxs0.copy()

    class NpProxy:
        random = RandomProxy()

        def __getattr__(self, k):
            return getattr(np, k)

    def shuffle(arr):
        out = real_shuffle(arr)
        rec["shuffled"].append(out.copy())
        return out

    def backward(self, *a_, **k_):
        rec["losses"].append(float(self.item()))
        return real_backward(self, *a_, **k_)

    rm.np, rm.shuffle_rows, torch.Tensor.backward = NpProxy(), shuffle, backward
    try:
        with contextlib.redirect_stderr(io.StringIO()):          # tqdm bar
            mpc.train(s, a, random=True, next_obs=s2, epochs=epochs)
    finally:
        rm.np, rm.shuffle_rows, torch.Tensor.backward = real_np, real_shuffle, real_backward
    return rec


def main():
    G = np.load(os.path.join(HERE, "mpc_golden.npz"))
    rng = np.random.RandomState(2024)
    np.random.seed(11)
    torch.manual_seed(11)
    from env.make_utils import register_env
    register_env("navigation2")

    n = 200
    s = rng.uniform([-45, -15], [-5, 15], (n, 2))
    a = rng.uniform(-1, 1, (n, 2))
    s2 = s + a + 0.05 * rng.randn(n, 2)
    out = {"data.s": s, "data.a": a, "data.s2": s2, "stride": np.array(STRIDE)}

    # ---- case "step": 32 rows, one epoch => exactly one optimiser step (MPC.py:266-296) ----
    mpc = build_reference_mpc(G)
    idxs = rng.randint(0, 32, size=(5, 32))
    rec = run_train(mpc, s[:32], a[:32], s2[:32], 1, idxs)
    assert len(rec["losses"]) == 1
    out["step.idxs"] = idxs
    out["step.loss"] = np.array(rec["losses"][0])
    out["step.mu"] = mpc.model.inputs_mu.detach().numpy().copy()
    out["step.sigma"] = mpc.model.inputs_sigma.detach().numpy().copy()
    for name in PARAMS:
        p = getattr(mpc.model, name)
        out["step.grad." + name] = subset(name, p.grad.detach().numpy())
        out["step.post." + name] = subset(name, p.detach().numpy())

    # ---- case "train": 200 rows, 2 epochs x ceil(200/32) = 14 steps, the last batch of an epoch has 8 rows ----
    mpc = build_reference_mpc(G)
    idxs = rng.randint(0, n, size=(5, n))
    rec = run_train(mpc, s, a, s2, 2, idxs)
    assert len(rec["losses"]) == 14 and len(rec["shuffled"]) == 2
    out["train.idxs"] = idxs
    out["train.shuffled"] = np.stack(rec["shuffled"])
    out["train.losses"] = np.array(rec["losses"])
    out["train.mu"] = mpc.model.inputs_mu.detach().numpy().copy()
    out["train.sigma"] = mpc.model.inputs_sigma.detach().numpy().copy()
    for name in PARAMS:
        out["train.post." + name] = subset(name, getattr(mpc.model, name).detach().numpy())
    assert mpc.train_in.shape == (n, 4) and mpc.has_been_trained
    out["train.train_in"] = mpc.train_in.astype(np.float32)
    out["train.train_targs"] = mpc.train_targs.astype(np.float32)

    np.savez_compressed(os.path.join(HERE, "mpc_train_golden.npz"), **out)
    print("wrote", len(out), "arrays; step loss", rec["losses"][0], "... last", rec["losses"][-1])


if __name__ == "__main__":
    main()
