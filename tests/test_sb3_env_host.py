"""CPU: the SB3-shaped env imports without SB3 / gymnasium / a GPU and offers SB3's VecEnv protocol (reference caller:
rl/train_ppo_sb3.py:49-141,246)."""
import inspect

from pednstream_amd import sb3_env


def test_vec_env_protocol_is_there():
    cls = sb3_env.PedNetSB3VecEnv
    for name in ("reset", "step_async", "step_wait", "step", "close", "get_attr", "set_attr", "env_method", "env_is_wrapped", "seed",
                 "set_options", "render", "get_images"):
        assert callable(getattr(cls, name)), name
    sig = inspect.signature(cls.__init__)
    assert {"dataset", "n_envs", "randomize", "normalize_obs"} <= set(sig.parameters)
    assert list(inspect.signature(cls.step_wait).parameters) == ["self"]
