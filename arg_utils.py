"""Command-line surface of `python -m rrl_main` -- flag-for-flag the reference's
arg_utils.get_args (arg_utils.py:8-257): same 55 flags, names, types, defaults and the same
argparse quirks (`--eval` / `--automatic_entropy_tuning` are `type=bool`, so any non-empty
string is True; prefix matching lets scripts write `--lambda 1000`).  The three flags in
ADDITIVE are new and default to the reference's behaviour.
"""
import argparse

F, I, S = float, int, None
# (flags, type-or-'store_true'-or-'append2', default, help)
REFERENCE_FLAGS = [
    # global
    (("--env-name",), S, "maze", "Gym environment (default: maze)"),
    (("--logdir",), S, "runs", "exterior log directory"),
    (("--logdir_suffix",), S, "", "log directory suffix"),
    (("--cuda",), "store_true", None, "run on the GPU (ROCm keeps the 'cuda' device name)"),
    (("--cnn",), "store_true", None, "visual observations (out of scope here)"),
    (("--lr",), F, 0.0003, "learning rate"),
    (("--updates_per_step",), I, 1, "model updates per simulator step"),
    (("--start_steps",), I, 100, "steps sampling random actions"),
    (("--target_update_interval",), I, 1, "value target update per no. of updates per step"),
    # task policy (SAC)
    (("--policy",), S, "Gaussian", "Gaussian | Deterministic"),
    (("--eval",), bool, True, "evaluate the policy every 10 episodes"),
    (("--gamma",), F, 0.99, "discount factor for reward"),
    (("--tau",), F, 0.005, "target smoothing coefficient"),
    (("--alpha",), F, 0.2, "entropy temperature"),
    (("--automatic_entropy_tuning",), bool, False, "automatically adjust alpha"),
    (("--seed",), I, 123456, "random seed"),
    (("--batch_size",), I, 256, "batch size"),
    (("--num_steps",), I, 1000000, "maximum number of steps"),
    (("--num_eps",), I, 1000000, "maximum number of episodes"),
    (("--hidden_size",), I, 256, "hidden size"),
    (("--replay_size",), I, 1000000, "size of replay buffer"),
    (("--task_demos",), "store_true", None, "use task demos to pretrain the task critic"),
    (("--num_task_transitions",), I, 10000000, "number of task transitions"),
    (("--critic_pretraining_steps",), I, 3000, "gradient steps for critic pretraining"),
    # Q risk
    (("--pos_fraction",), F, -1, "fraction of positive examples for critic training"),
    (("--gamma_safe",), F, 0.5, "discount factor for constraints"),
    (("--eps_safe",), F, 0.1, "Qrisk threshold"),
    (("--tau_safe",), F, 0.0002, "Qrisk target smoothing coefficient"),
    (("--safe_replay_size",), I, 1000000, "size of replay buffer for Qrisk"),
    (("--num_unsafe_transitions",), I, 10000, "number of unsafe transitions"),
    (("--critic_safe_pretraining_steps",), I, 10000, "gradient steps for Qrisk pretraining"),
    # recovery
    (("--use_recovery",), "store_true", None, "use recovery policy"),
    (("--MF_recovery",), "store_true", None, "model free recovery policy"),
    (("--Q_sampling_recovery",), "store_true", None, "sample actions over Qrisk for recovery"),
    (("-ca", "--ctrl_arg"), "append2", [], "controller arguments (parsed, inert -- as in the reference)"),
    (("-o", "--override"), "append2", [], "config overrides (parsed, inert -- as in the reference)"),
    (("--recovery_policy_update_freq",), I, 1, "model updated every this many episodes"),
    (("--vismpc_recovery",), "store_true", None, "visual model-based recovery (out of scope here)"),
    (("--load_vismpc",), "store_true", None, "load pre-trained visual dynamics model"),
    (("--model_fname",), S, "image_maze_dynamics", "path to pre-trained visual dynamics model"),
    (("--beta",), F, 10, "beta for the visual dynamics VAE"),
    # ablations
    (("--disable_offline_updates",), "store_true", None, "only train Qrisk online"),
    (("--disable_online_updates",), "store_true", None, "only train Qrisk on offline data"),
    (("--disable_action_relabeling",), "store_true", None, "train task policy on recovery policy actions"),
    (("--add_both_transitions",), "store_true", None, "use both task and recovery transitions"),
    # comparisons
    (("--constraint_reward_penalty",), F, 0, "reward penalty when a constraint is violated"),
    (("--DGD_constraints",), "store_true", None, "dual gradient descent on task reward + constraints"),
    (("--use_constraint_sampling",), "store_true", None, "sample actions with task policy, filter with Qrisk"),
    (("--nu",), F, 0.01, "penalty term in Lagrangian objective"),
    (("--update_nu",), "store_true", None, "update Lagrangian penalty term"),
    (("--nu_schedule",), "store_true", None, "linear schedule for nu"),
    (("--nu_start",), F, 1e3, "start value for nu"),
    (("--nu_end",), F, 0, "end value for nu"),
    (("--RCPO",), "store_true", None, "use RCPO"),
    (("--lambda_RCPO",), F, 0.01, "penalty term for RCPO"),
]

ADDITIVE = [
    (("--num_envs",), I, 1, "independent envs advanced in lock-step on the GPU (1 = reference loop)"),
    (("--log_every",), I, 0, "vector steps between metric aggregations across ranks (0 = per episode)"),
    (("--mb_dynamics",), S, "model", "CEM rollouts through the learned ensemble ('model', reference) "
                                     "or the env kernels ('env', extension)"),
    (("--checkpoint_every",), I, 0, "lock-step loop: write <logdir>/checkpoint.pt every this many log intervals "
                                    "(0 = only at the end)"),
    (("--dp_mode",), S, "replicas", "multi-GPU mode under torchrun: 'replicas' = one independent seed per GPU "
                                    "(reference semantics), 'env_shard' = ONE learner, envs and replay split over "
                                    "the GPUs, gradients averaged with RCCL before every optimiser step"),
    (("--no_fast_path",), "store_true", None, "SAC / Q_risk updates through autograd instead of the fused kernels"),
    (("--plan_precision",), S, "", "hidden layers of the fused planner kernel: 'f32' = f32 MFMA (exact f32 products), "
                                   "'f16x3' = three f16 MFMA products of hi/lo splits (operands to 2^-22 relative, 3e-8 absolute for "
                                   "values below 0.06; 2.7x faster); "
                                   "default: f32 unless RRL_PLAN_F16X3=1"),
    (("--graph_iterations",), I, 4, "Lock-step driver: iterations captured per hipGraph (the steady state is replayed in graphs of "
     "this many iterations wherever that many fit before the next log point; 1 = one iteration per graph)"),
    (("--plan_seed",), I, 0, "Philox key of the model-based recovery controller's own streams (CEM samples, particle noise of "
                             "the planner kernel); everything else keeps following --seed"),
    (("--resume",), S, "", "checkpoint.pt to continue from (lock-step loop; skips pre-training)"),
    (("--seeds_per_gpu",), I, 1, "lock-step loop: run this many independent seeds (seed, seed + 1, ...: own envs, replay "
                                 "rings, networks, Philox keys -- the reference's seed loop, scripts/navigation1.sh:4-8) on "
                                 "ONE GPU, sharing every launch of the iteration (recovery_rl_amd/packed.py)"),
    (("--info_envs",), I, 0, "lock-step loop: keep the reference's per-step info dicts (run_stats.pkl `train_stats`, "
                             "experiment.py:421,540-543) for the first K envs, so that plotting/plot_runs.py reads the run "
                             "unchanged (0 = per-episode table only)"),
    (("--no_pin_demos",), "store_true", None, "lock-step loop: let the safety buffer's ring overwrite the offline constraint "
                                              "demonstrations (default: they are pinned, as the one-env reference never "
                                              "wraps its 1e6-row ring within a run)"),
    (("--keep_replay_size",), "store_true", None, "lock-step loop: keep --replay_size / --safe_replay_size as given even when "
                                                  "--num_steps exceeds them (default at --num_envs > 1: both buffers are sized "
                                                  "to hold the whole run, as the reference's defaults do: replay_size = "
                                                  "num_steps = 1e6)"),
    (("--keep_plan_warm_start",), "store_true", None, "lock-step loop, model-based recovery: keep an env's CEM warm start "
                                                      "(MPC.prev_sol) across its episodes, as the one-env reference does "
                                                      "(default at --num_envs > 1: an episode's first plan starts from the "
                                                      "mid-point action sequence, MPC.py:174 -- vectorisation rule 5)"),
    (("--demo_share",), F, -1.0, "lock-step loop: share of every Q_risk batch drawn from the pinned constraint "
                                 "demonstrations, the rest from the online rows (0 = one uniform draw over the ring; "
                                 "default -1: 0.5 with --num_envs > 1 and pinned demonstrations -- the share a one-env "
                                 "reference run sees from its first to its 400th episode, experiment.py:278-286,438-448 -- "
                                 "and 0 otherwise; ignored with --pos_fraction)"),
]


def build_parser():
    parser = argparse.ArgumentParser(description='Recovery RL Arguments')
    for flags, kind, default, help_ in REFERENCE_FLAGS + ADDITIVE:
        if kind == "store_true":
            parser.add_argument(*flags, action='store_true', help=help_)
        elif kind == "append2":
            parser.add_argument(*flags, action='append', nargs=2, default=default, help=help_)
        elif kind is None:
            parser.add_argument(*flags, default=default, help=help_)
        else:
            parser.add_argument(*flags, type=kind, default=default, help=help_)
    return parser


def get_args(argv=None):
    return build_parser().parse_args(argv)
