"""Inputs of the planner's production-shape known-answer test (Q_risk hidden 256 -- arg_utils.py:77 default --, PETS ensemble
5 x 200, 400 candidates x 20 particles x 5 steps: config/navigation2.py CEM settings): everything is re-created from seeded
numpy streams, so `mpc_golden_256.npz` holds only what the REFERENCE produced from these inputs (gen_mpc_golden_256.py).

Used by the fixture generator (imports the reference) and by tests/test_plan_kat256_gpu.py (this stack's rrl_plan_cost).
"""
import numpy as np

from kat256_inputs import _rs, weights as linear_weights

HQ, HE, NETS, NPART, POP, PLAN_HOR = 256, 200, 5, 20, 400, 5
ARGV = ["--use_recovery", "--gamma_safe", "0.65", "--eps_safe", "0.2"]            # scripts/navigation2.sh:14 (model-based line)
ACTION_GAIN = 6.0        # first-layer action columns x 6: candidates then differ in cost by ~0.1 (Xavier weights alone make
                         # Q_risk a function of the O(30) observation only, and every candidate costs the same)
# planning problems: start of an episode, beside the obstacle (navigation2.py:41: x in [-30, -20], |y| <= 7.5), far corner
CUR_OBS = np.array([[-33.0, 2.5], [-31.2, -6.9], [-45.0, 9.0]], dtype=np.float64)
Q_SAMPLE_ROWS = 96       # particle rows per (problem, step) whose Q_risk value the fixture records


def qrisk_weights(state_dict):
    """Twin Q_risk (QNetworkConstraint, model.py:232-267): Xavier-uniform, biases in [-0.2, 0.2]; the action columns of the
    two first layers x ACTION_GAIN."""
    out = linear_weights(state_dict, "plan.qrisk")
    for k in ("linear1.weight", "linear4.weight"):
        out[k] = out[k].copy()
        out[k][:, 2:] *= ACTION_GAIN
    return out


def ensemble_weights():
    """PtModel parameters (config/navigation2.py:23-48): weights N(0, 1 / (4 fan_in)) (the scale get_affine_params draws
    at, config/utils.py), last layer x 3, biases N(0, 0.1); logvar bounds at their initial values."""
    out = {}
    dims = [(4, HE), (HE, HE), (HE, HE), (HE, 4)]
    for i, (din, dout) in enumerate(dims):
        rs = _rs("plan.ens", i)
        w = rs.randn(NETS, din, dout) / (2.0 * np.sqrt(din))
        if i == 3:
            w = w * 3.0
        out["lin%d_w" % i] = w.astype(np.float32)
        out["lin%d_b" % i] = (0.1 * rs.randn(NETS, 1, dout)).astype(np.float32)
    return out


def stats_data():
    """Rows the input statistics are fitted on (PtModel.fit_input_stats): Navigation2-like (obs, action) rows."""
    rs = _rs("plan.stats")
    return rs.randn(2000, 4) * [12.0, 6.0, 0.6, 0.6] + [-35.0, 0.5, 0.0, 0.0]


def candidates():
    """ac_seqs [M, POP, PLAN_HOR * 2] in [-1, 1] (what CEM hands _compile_cost, MPC.py:375), float32."""
    return _rs("plan.acs").uniform(-1, 1, (len(CUR_OBS), POP, PLAN_HOR * 2)).astype(np.float32)


def noise():
    """Particle noise [PLAN_HOR, M, POP * NPART, 2], row = c * NPART + p (the flat row order of MPC.py:389-399)."""
    return _rs("plan.noise").randn(PLAN_HOR, len(CUR_OBS), POP * NPART, 2).astype(np.float32)


def q_sample_rows():
    return np.sort(_rs("plan.rows").choice(POP * NPART, Q_SAMPLE_ROWS, replace=False))
