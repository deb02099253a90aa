"""Navigation1 / Navigation2 on the MI355X: host-side mirror of env/navigation1.py and
env/navigation2.py over the C-ABI kernels rrl_nav_step / rrl_nav_reset / rrl_nav_offline.

`NavigationVecEnv` advances `num_envs` independent episodes in lock-step; all state lives
in HBM as structure-of-arrays tensors and `step` is one kernel launch.  `Navigation1` /
`Navigation2` are the single-env objects with the reference's gym protocol
(reset() -> obs, step(a) -> (obs, reward, done, info); env/navigation1.py:71-97).
"""
import numpy as np
import torch

from .. import _lib
from ..spaces import Box
from .status import StatusWordMixin

ENV_KIND = {"navigation1": 0, "navigation2": 1}

# constants of env/navigation1.py:27-36 (identical in navigation2.py)
START_STATE = (-50.0, 0.0)
GOAL_STATE = (0.0, 0.0)
MAX_FORCE = 1.0
HORIZON = 100
NOISE_SCALE = 0.05


class NavigationVecEnv(StatusWordMixin):
    """Batched Navigation1/2.

    step(action[N,2] f32) -> (obs[N,2], reward[N], done[N] bool, info) where `info` holds
    per-env tensors with the reference's info keys (navigation1.py:82-89) plus
    `ep_done` (done or horizon, experiment.py:435).  The returned tensors are persistent
    buffers overwritten by the next call.  With auto_reset (default) finished envs restart
    inside the same launch: `obs` is then the first observation of the next episode while
    info["next_state"] is the terminal s' that goes to replay (experiment.py:439,449).
    """

    def __init__(self, env_name, num_envs, device="cuda", seed=0, horizon=HORIZON,
                 auto_reset=True):
        self.env_name = env_name
        self.kind = ENV_KIND[env_name]
        self.device = _lib.require_gpu(device)
        self.lib = _lib.load()
        self.num_envs = int(num_envs)
        self.seed_value = int(seed) & 0xFFFFFFFFFFFFFFFF
        self.horizon = self._max_episode_steps = int(horizon)
        self.auto_reset = bool(auto_reset)
        self.action_space = Box(-np.ones(2) * MAX_FORCE, np.ones(2) * MAX_FORCE)
        self.observation_space = Box(-np.ones(2) * float("inf"), np.ones(2) * float("inf"))
        self.goal = GOAL_STATE
        self.transition_function = self.get_offline_data
        n, dev = self.num_envs, self.device
        self.pos = torch.zeros(n, 2, dtype=torch.float64, device=dev)
        self.t = torch.zeros(n, dtype=torch.int32, device=dev)
        self.obs = torch.zeros(n, 2, dtype=torch.float32, device=dev)
        self.prev_obs = torch.zeros(n, 2, dtype=torch.float32, device=dev)
        self.next_obs = torch.zeros(n, 2, dtype=torch.float32, device=dev)
        self.reward = torch.zeros(n, dtype=torch.float32, device=dev)
        self.action_clipped = torch.zeros(n, 2, dtype=torch.float32, device=dev)
        self._flags = torch.zeros(4, n, dtype=torch.uint8, device=dev)
        self.done, self.constraint, self.success, self.ep_done = self._flags.unbind(0)
        self._init_status()
        # RNG tick {tick, ticket}: lives on the device so that captured graphs advance it
        self.tick = torch.zeros(2, dtype=torch.int64, device=dev)

    # -- gym-style helpers --------------------------------------------------------------
    def seed(self, seed=None):
        if seed is not None:
            self.seed_value = int(seed) & 0xFFFFFFFFFFFFFFFF
        return [seed]

    def sample_actions(self, generator=None):
        """Uniform random actions in the action box (action_space.sample per env)."""
        return torch.rand(self.num_envs, 2, device=self.device, generator=generator) * 2 * MAX_FORCE - MAX_FORCE

    # -- protocol ---------------------------------------------------------------------------
    def reset(self, mask=None, noise=None):
        """All envs (or those with mask != 0) restart at START_STATE + N(0, I)."""
        self.use_arrays()
        rc = self.lib.rrl_nav_reset(self.kind, self.num_envs, _lib.ptr(self.pos), _lib.ptr(self.obs),
                                    _lib.ptr(self.t), _lib.ptr(mask), _lib.ptr(noise),
                                    self.seed_value, 0, _lib.ptr(self.tick), _lib.current_stream())
        _lib.check(rc, "rrl_nav_reset")
        # reset and step share the tick; bump it so the next step draws fresh reset noise
        rc = self.lib.rrl_counter_add(_lib.ptr(self.tick), 1, _lib.current_stream())
        _lib.check(rc, "rrl_counter_add")
        return self.obs

    def step(self, action, noise=None):
        assert action.dtype == torch.float32 and action.is_contiguous()
        assert action.shape == (self.num_envs, 2)
        self.use_arrays()
        self.prev_obs.copy_(self.obs)
        rc = self.lib.rrl_nav_step(
            self.kind, self.num_envs, _lib.ptr(self.pos), _lib.ptr(action), _lib.ptr(noise),
            self.seed_value, 0, _lib.ptr(self.tick), 1, _lib.ptr(self.next_obs), _lib.ptr(self.obs),
            _lib.ptr(self.reward), _lib.ptr(self.done), _lib.ptr(self.constraint),
            _lib.ptr(self.success), _lib.ptr(self.ep_done), _lib.ptr(self.t), self.horizon,
            int(self.auto_reset), _lib.current_stream())
        _lib.check(rc, "rrl_nav_step")
        torch.clamp(action, -MAX_FORCE, MAX_FORCE, out=self.action_clipped)
        info = {"constraint": self.constraint, "reward": self.reward, "state": self.prev_obs,
                "next_state": self.next_obs, "action": self.action_clipped,
                "success": self.success, "ep_done": self.ep_done}
        return self.obs, self.reward, self.done, info

    def get_offline_data(self, num_transitions, task_demos=False, seed=None):
        """Constraint demonstrations (navigation1.py:133-164 / navigation2.py:133-243) as five
        device tensors (state, action, constraint, next_state, mask)."""
        if task_demos:
            raise NotImplementedError("the navigation envs define no task demos (reference returns "
                                      "only constraint transitions)")
        return offline_data(self.env_name, num_transitions,
                            self.seed_value if seed is None else seed, self.device)


def offline_data(env_name, num_transitions, seed, device="cuda"):
    dev = _lib.require_gpu(device)
    lib = _lib.load()
    kind = ENV_KIND[env_name]
    n_roll = lib.rrl_nav_offline_rollouts(kind, int(num_transitions))
    cap = max(10 * n_roll, 1)
    s = torch.empty(cap, 2, dtype=torch.float32, device=dev)
    a = torch.empty(cap, 2, dtype=torch.float32, device=dev)
    c = torch.empty(cap, dtype=torch.float32, device=dev)
    s2 = torch.empty(cap, 2, dtype=torch.float32, device=dev)
    m = torch.empty(cap, dtype=torch.float32, device=dev)
    count = torch.zeros(1, dtype=torch.int64, device=dev)
    scratch = torch.zeros(n_roll + 1, dtype=torch.int32, device=dev)
    rc = lib.rrl_nav_offline(kind, int(num_transitions), int(seed) & 0xFFFFFFFFFFFFFFFF,
                             _lib.ptr(s), _lib.ptr(a), _lib.ptr(c), _lib.ptr(s2), _lib.ptr(m), cap,
                             _lib.ptr(count), _lib.ptr(scratch), _lib.current_stream())
    _lib.check(rc, "rrl_nav_offline")
    w = int(count.item())
    return s[:w], a[:w], c[:w], s2[:w], m[:w]


class _SingleNavigation:
    """One env with the reference's numpy protocol (env/navigation1.py:55-97)."""
    ENV_NAME = None

    def __init__(self, device="cuda", seed=0):
        self._vec = NavigationVecEnv(self.ENV_NAME, 1, device=device, seed=seed, auto_reset=False)
        self.action_space = self._vec.action_space
        self.observation_space = self._vec.observation_space
        self._max_episode_steps = self.horizon = HORIZON
        self.goal = list(GOAL_STATE)
        self.transition_function = self.get_offline_data
        self.state = None
        self.time = 0
        self.done = False

    def seed(self, seed=None):
        return self._vec.seed(seed)

    def reset(self):
        self._vec.reset()
        self.state = self._vec.pos[0].cpu().numpy().copy()
        self.time = 0
        self.done = False
        return self.state

    def step(self, a):
        act = torch.as_tensor(np.asarray(a, dtype=np.float32).reshape(1, 2), device=self._vec.device)
        old_state = self.state.copy()
        _, reward, done, info = self._vec.step(act.contiguous())
        self.state = self._vec.pos[0].cpu().numpy().copy()
        self.time += 1
        self.done = bool(done[0].item())
        cost = float(reward[0].item())
        return self.state, cost, self.done, {
            "constraint": int(info["constraint"][0].item()),
            "reward": cost,
            "state": old_state,
            "next_state": self.state,
            "action": info["action"][0].cpu().numpy(),
            "success": bool(info["success"][0].item()),
        }

    def get_offline_data(self, num_transitions, task_demos=False, save_rollouts=False):
        s, a, c, s2, m = self._vec.get_offline_data(num_transitions, task_demos)
        s, a, c, s2, m = (x.cpu().numpy() for x in (s, a, c, s2, m))
        return [(s[i], a[i], int(c[i]), s2[i], bool(m[i])) for i in range(len(c))]


class Navigation1(_SingleNavigation):
    ENV_NAME = "navigation1"


class Navigation2(_SingleNavigation):
    ENV_NAME = "navigation2"
