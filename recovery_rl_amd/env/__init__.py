from .make_utils import ENV_CLASS, ENV_ID, make_env, make_vec_env, register_env  # noqa: F401
