"""What the example scripts share: the compat aliases for the reference's import paths and a one-line summary of a finished run."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pednstream_amd.compat as compat  # noqa: E402

compat.install()


def summary(network_env, last):
    """Busiest link at time index `last`, totals over the physical links, and the engine's error flags raised as exceptions."""
    network_env.engine().check_errors()
    links = [l for l in network_env.links.values() if not l.is_virtual]
    total_in = sum(float(l.cumulative_inflow[last]) for l in links)
    on_net = sum(float(l.num_pedestrians[last]) for l in links)
    busiest = max(links, key=lambda l: float(l.num_pedestrians[last]))
    return (f"cumulative inflow over {len(links)} links at t={last}: {total_in:.0f}; pedestrians on the network: {on_net:.0f}; busiest link "
            f"{busiest.link_id}: {float(busiest.num_pedestrians[last]):.0f} pedestrians, density {float(busiest.density[last]):.3f}")


def save(network_env, name):
    """The reference's examples end with OutputHandler(...).save_network_state(network) (then a NetworkVisualizer animation, which is
    plotting and not provided)."""
    from handlers.output_handler import OutputHandler      # reference import path (compat)

    out = OutputHandler(base_dir=os.path.join(ROOT, "outputs"), simulation_dir=name)
    out.save_network_state(network_env)
    return out.simulation_dir
