"""pednstream_amd -- MI355X-native engine for PedNStream's per-timestep ``network_loading`` hot path.

Public surface (mirrors the reference's for this path):
    NetworkEnvGenerator   scenario directory -> Network            (reference src/utils/env_loader.py)
    Network               network_loading(t), links, nodes, ...    (reference src/LTM/network.py)
    load_config           YAML/JSON scenario -> config dict         (reference src/utils/config.py)
"""
from .config import load_config
from .env_loader import NetworkEnvGenerator
from .network import Network

__all__ = ["NetworkEnvGenerator", "Network", "load_config"]
__version__ = "0.1.0"
