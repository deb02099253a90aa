"""Maze on the MI355X: host-side mirror of env/maze.py over rrl_maze_step / rrl_maze_reset /
rrl_maze_offline.

PARITY UNPINNED: the reference integrates MuJoCo 1.50 (`mujoco_py`, third-party, not in the
reference tree or this image).  Control flow, rewards, termination, reset ranges, wall geometry
and the scripted expert follow env/maze.py; the physics is the kinematic surrogate of
DESIGN.md section 6.  This env matches the reference's *task definition*, not MuJoCo trajectories.
"""
import numpy as np
import torch

from .. import _lib
from ..spaces import Box
from .status import StatusWordMixin

HORIZON = 100          # env/maze.py:16
MAX_FORCE = 0.1        # env/maze.py:17
GOAL_THRESH = 3e-2     # env/maze.py:19
GOAL = (0.25, 0.0)     # env/maze.py:135-137
RESET_MODES = {'h': 0, 'e': 1, 'm': 2, None: 3}


class MazeVecEnv(StatusWordMixin):
    """Batched MazeNavigation; same step contract as NavigationVecEnv.  `done` already includes
    the env's own horizon (env/maze.py:153), so the bootstrap mask is 0 on time-outs, as in the
    reference."""

    def __init__(self, env_name="maze", num_envs=1, device="cuda", seed=0, horizon=HORIZON,
                 auto_reset=True):
        self.env_name = "maze"
        self.device = _lib.require_gpu(device)
        self.lib = _lib.load()
        self.num_envs = int(num_envs)
        self.seed_value = int(seed) & 0xFFFFFFFFFFFFFFFF
        self.horizon = self._max_episode_steps = int(horizon)
        self.auto_reset = bool(auto_reset)
        self.action_space = Box(-MAX_FORCE * np.ones(2), MAX_FORCE * np.ones(2))
        self.observation_space = Box(-0.3, 0.3, shape=(2,))
        self.goal = np.array(GOAL)
        self.gain = 1.05
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
        self.tick = torch.zeros(2, dtype=torch.int64, device=dev)

    def seed(self, seed=None):
        if seed is not None:
            self.seed_value = int(seed) & 0xFFFFFFFFFFFFFFFF
        return [seed]

    def sample_actions(self, generator=None):
        return (torch.rand(self.num_envs, 2, device=self.device, generator=generator) * 2 - 1) * MAX_FORCE

    def reset(self, difficulty='h', check_constraint=True, pos=(), mask=None):
        """env/maze.py:184-213."""
        self.use_arrays()
        if len(pos):
            self.pos[:] = torch.as_tensor(pos, dtype=torch.float64, device=self.device)
            self.t.zero_()
            self.obs.copy_(self.pos.to(torch.float32))
            return self.obs
        rc = self.lib.rrl_maze_reset(self.num_envs, _lib.ptr(self.pos), _lib.ptr(self.obs), _lib.ptr(self.t),
                                     _lib.ptr(mask), RESET_MODES[difficulty], int(check_constraint),
                                     self.seed_value, 0, _lib.ptr(self.tick), _lib.current_stream())
        _lib.check(rc, "rrl_maze_reset")
        _lib.check(self.lib.rrl_counter_add(_lib.ptr(self.tick), 1, _lib.current_stream()), "rrl_counter_add")
        return self.obs

    def step(self, action):
        assert action.dtype == torch.float32 and action.is_contiguous()
        assert action.shape == (self.num_envs, 2)
        self.use_arrays()
        self.prev_obs.copy_(self.obs)
        rc = self.lib.rrl_maze_step(
            self.num_envs, _lib.ptr(self.pos), _lib.ptr(action), self.seed_value, 0, _lib.ptr(self.tick), 1,
            _lib.ptr(self.next_obs), _lib.ptr(self.obs), _lib.ptr(self.reward), _lib.ptr(self.done),
            _lib.ptr(self.constraint), _lib.ptr(self.success), _lib.ptr(self.ep_done), _lib.ptr(self.t),
            self.horizon, int(self.auto_reset), _lib.current_stream())
        _lib.check(rc, "rrl_maze_step")
        torch.clamp(action, -MAX_FORCE, MAX_FORCE, out=self.action_clipped)
        info = {"constraint": self.constraint, "reward": self.reward, "state": self.prev_obs,
                "next_state": self.next_obs, "action": self.action_clipped, "success": self.success,
                "ep_done": self.ep_done}
        return self.obs, self.reward, self.done, info

    def expert_action(self):
        """env/maze.py:222-232 for every env."""
        x = self.pos[:, 0:1]
        t1 = torch.tensor([-0.15, -0.125], dtype=torch.float64, device=self.device)
        t2 = torch.tensor([0.15, 0.125], dtype=torch.float64, device=self.device)
        t3 = torch.tensor(GOAL, dtype=torch.float64, device=self.device)
        target = torch.where(x <= -0.151, t1, torch.where(x <= 0.149, t2, t3))
        return (self.gain * (target - self.pos)).to(torch.float32)

    def get_offline_data(self, num_transitions, task_demos=False, seed=None):
        """Constraint demonstrations (generator of env/maze.py:34-107; the reference's driver loads
        them from demos/maze/constraint_demos.pkl, experiment.py:195-199, a file not in the tree)."""
        if task_demos:
            raise NotImplementedError("the maze env defines no task demos")
        return offline_data(num_transitions, self.seed_value if seed is None else seed, self.device)


def offline_data(num_transitions, seed, device="cuda"):
    dev = _lib.require_gpu(device)
    lib = _lib.load()
    cap = max(2 * (int(num_transitions) // 2), 1)
    s = torch.empty(cap, 2, dtype=torch.float32, device=dev)
    a = torch.empty(cap, 2, dtype=torch.float32, device=dev)
    c = torch.empty(cap, dtype=torch.float32, device=dev)
    s2 = torch.empty(cap, 2, dtype=torch.float32, device=dev)
    m = torch.empty(cap, dtype=torch.float32, device=dev)
    rc = lib.rrl_maze_offline(int(num_transitions), int(seed) & 0xFFFFFFFFFFFFFFFF, _lib.ptr(s), _lib.ptr(a),
                              _lib.ptr(c), _lib.ptr(s2), _lib.ptr(m), cap, _lib.current_stream())
    _lib.check(rc, "rrl_maze_offline")
    w = 2 * (int(num_transitions) // 2)
    return s[:w], a[:w], c[:w], s2[:w], m[:w]


class MazeNavigation:
    """One env with the reference's numpy protocol (env/maze.py:110-232)."""

    def __init__(self, device="cuda", seed=0):
        self._vec = MazeVecEnv("maze", 1, device=device, seed=seed, auto_reset=False)
        self.action_space = self._vec.action_space
        self.observation_space = self._vec.observation_space
        self.horizon = self._max_episode_steps = HORIZON
        self.goal = self._vec.goal
        self.gain = 1.05
        self.transition_function = self.get_offline_data
        self.steps = 0
        self.done = False
        self.dense_reward = True

    def seed(self, seed=None):
        return self._vec.seed(seed)

    def _get_obs(self, images=False):
        if images:
            raise NotImplementedError("image rendering needs MuJoCo + GL (out of scope)")
        return self._vec.pos[0].cpu().numpy().copy()

    def reset(self, difficulty='h', check_constraint=True, pos=()):
        self._vec.reset(difficulty, check_constraint, pos)
        self.steps = 0
        return self._get_obs()

    def step(self, action):
        act = torch.as_tensor(np.asarray(action, dtype=np.float32).reshape(1, 2), device=self._vec.device)
        cur_obs = self._get_obs()
        _, reward, done, info = self._vec.step(act.contiguous())
        obs = self._get_obs()
        self.steps += 1
        self.done = bool(done[0].item())
        r = float(reward[0].item())
        return obs, r, self.done, {
            "constraint": int(info["constraint"][0].item()), "reward": r, "state": cur_obs,
            "next_state": obs, "action": info["action"][0].cpu().numpy(),
            "success": bool(info["success"][0].item())}

    def get_distance_score(self):
        p = self._get_obs()
        return float(np.sqrt(np.mean((self.goal - p) ** 2)))

    def expert_action(self):
        return self._vec.expert_action()[0].cpu().numpy()

    def get_offline_data(self, num_transitions, images=False, save_rollouts=False):
        s, a, c, s2, m = (x.cpu().numpy() for x in self._vec.get_offline_data(num_transitions))
        return [(s[i], a[i], int(c[i]), s2[i], bool(m[i])) for i in range(len(c))]
