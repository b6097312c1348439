"""Drop-in aliases for the reference's import paths.

The reference's callers import ``src.utils.env_loader.NetworkEnvGenerator``, ``src.LTM.network.Network``,
``src.utils.config.load_config``, ``handlers.output_handler.OutputHandler`` and ``rl.PedNetParallelEnv`` (e.g.
examples/six_node.py:8-10, rl/rl_example.py).  ``install()`` registers modules
of those names that resolve to this package, so such a script runs on the MI355X engine unchanged:

    import pednstream_amd.compat as compat; compat.install()
    from src.utils.env_loader import NetworkEnvGenerator      # -> pednstream_amd.env_loader
"""
import sys
import types


def install(force: bool = False):
    from . import config, env_loader, network, od_manager, output_handler, path_finder, rl_env

    def mod(name, **attrs):
        if name in sys.modules and not force:
            m = sys.modules[name]
        else:
            m = types.ModuleType(name)
            m.__path__ = []          # behave as a package so that submodule imports resolve through sys.modules
            sys.modules[name] = m
        for k, v in attrs.items():
            setattr(m, k, v)
        return m

    src = mod("src")
    ltm = mod("src.LTM", Network=network.Network)
    utils = mod("src.utils")
    src.LTM, src.utils = ltm, utils
    ltm.network = mod("src.LTM.network", Network=network.Network)
    ltm.od_manager = mod("src.LTM.od_manager", ODManager=od_manager.ODManager, DemandGenerator=od_manager.DemandGenerator)
    ltm.path_finder = mod("src.LTM.path_finder", PathFinder=path_finder.PathFinder)
    ltm.link = mod("src.LTM.link", Link=network.LinkView, BaseLink=network.BaseLinkView, Separator=network.LinkView)
    ltm.node = mod("src.LTM.node", Node=network.NodeView)
    utils.env_loader = mod("src.utils.env_loader", NetworkEnvGenerator=env_loader.NetworkEnvGenerator)
    utils.config = mod("src.utils.config", load_config=config.load_config, validate_config=config.validate_config)
    # `from handlers.output_handler import OutputHandler` (examples/*.py) and `from rl import PedNetParallelEnv`
    # (rl/rl_example.py, rl/train_*.py); the plotting module src.utils.visualizer is not provided
    handlers = mod("handlers")
    handlers.output_handler = mod("handlers.output_handler", OutputHandler=output_handler.OutputHandler)
    rl = mod("rl", PedNetParallelEnv=rl_env.PedNetParallelEnv)
    rl.pz_pednet_env = mod("rl.pz_pednet_env", PedNetParallelEnv=rl_env.PedNetParallelEnv)
    rl.discovery = mod("rl.discovery", AgentManager=rl_env.AgentManager)
    return src
