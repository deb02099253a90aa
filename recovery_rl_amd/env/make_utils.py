"""Env registry -- the drop-in boundary of env/make_utils.py:4-31.

`register_env(name)` / `make_env(name)` keep the reference's names and error behaviour
(unknown name -> KeyError).  `make_vec_env(name, num_envs, device, seed)` is the additive
entry point for the batched MI355X path."""

ENV_ID = {  # env/make_utils.py:4-11 (the image / extraction envs are out of scope)
    'navigation1': 'Navigation-v0',
    'navigation2': 'Navigation-v1',
    'maze': 'Maze-v0',
}

ENV_CLASS = {  # env/make_utils.py:13-20
    'navigation1': 'Navigation1',
    'navigation2': 'Navigation2',
    'maze': 'MazeNavigation',
}

_REGISTERED = {}


def _entry(env_name):
    from . import navigation
    table = {'navigation1': (navigation.Navigation1, navigation.NavigationVecEnv),
             'navigation2': (navigation.Navigation2, navigation.NavigationVecEnv)}
    try:
        from . import maze
        table['maze'] = (maze.MazeNavigation, maze.MazeVecEnv)
    except ImportError:
        pass
    return table[env_name]


def register_env(env_name):
    env_id = ENV_ID[env_name]
    _REGISTERED[env_id] = env_name


def make_env(env_name, device='cuda', seed=0):
    """Single-env object with the reference's gym protocol (numpy in / numpy out)."""
    env_id = ENV_ID[env_name]
    if env_id not in _REGISTERED:
        raise KeyError("env %r is not registered; call register_env(%r) first" % (env_id, env_name))
    return _entry(env_name)[0](device=device, seed=seed)


def make_vec_env(env_name, num_envs, device='cuda', seed=0, **kw):
    """num_envs independent episodes advanced in lock-step on the GPU (tensors in / out)."""
    ENV_ID[env_name]
    return _entry(env_name)[1](env_name, num_envs, device=device, seed=seed, **kw)
