"""The reference's import paths resolve to this package after compat.install() (drop-in for examples/ and rl/)."""
import subprocess
import sys

CODE = r'''
import pednstream_amd.compat as compat
compat.install()
from src.utils.env_loader import NetworkEnvGenerator
from src.LTM.network import Network
from src.utils.config import load_config
from handlers.output_handler import OutputHandler
from rl import PedNetParallelEnv
from rl.discovery import AgentManager
import pednstream_amd, pednstream_amd.output_handler, pednstream_amd.rl_env
assert OutputHandler is pednstream_amd.output_handler.OutputHandler and PedNetParallelEnv is pednstream_amd.rl_env.PedNetParallelEnv
assert NetworkEnvGenerator is pednstream_amd.NetworkEnvGenerator and Network is pednstream_amd.Network
gen = NetworkEnvGenerator("data")
net = gen.create_network("od_flow_example", verbose=False)
assert gen.config["params"]["simulation_steps"] == 500          # examples/six_node.py:27 reads this
assert sorted(net.links)[0] == (0, 1) and net.links[(3, 5)].back_gate_width == 1
net.links[(3, 5)].back_gate_width -= 0.1                            # examples/six_node.py:30
assert abs(net.links[(5, 3)].front_gate_width - 0.9) < 1e-15
assert net.path_finder.od_paths == {(1, 5): [[1, 3, 5], [1, 3, 2, 4, 5]]}     # SURVEY 8c, measured on the reference
print("ok")
'''


def test_reference_import_paths_resolve():
    out = subprocess.run([sys.executable, "-c", CODE], capture_output=True, text=True, cwd=__import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
    assert out.returncode == 0 and out.stdout.strip() == "ok", out.stderr
