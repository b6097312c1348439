import time, sys, numpy as np
sys.path.insert(0, '.')
from pednstream_amd import NetworkEnvGenerator
for name in ("od_flow_example", "nine_intersections", "45_intersections", "delft", "melbourne"):
    gen = NetworkEnvGenerator("data")
    t0 = time.perf_counter(); net = gen.create_network(name, verbose=False); net.engine(); t1 = time.perf_counter()
    T = net.simulation_steps
    for t in range(1, T):
        net.network_loading(t)
    net.synchronize(); t2 = time.perf_counter()
    x = float(np.asarray(list(net.links.values())[0].cumulative_inflow)[T - 1])
    print(f"{name}: build+create {t1 - t0:.3f} s, {T - 1} x network_loading(t) {1e3 * (t2 - t1):.1f} ms ({1e6 * (t2 - t1) / (T - 1):.1f} us/step), {len(net.links) * (T - 1) / (t2 - t1):.3g} link-updates/s single replica")
    net.close()
