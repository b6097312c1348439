import ctypes, os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["PEDN_HIP_LIB"] = os.path.join(ROOT, "pednstream_amd/csrc/exp_PHASE.so")
from bench import replica_demand
from pednstream_amd import NetworkEnvGenerator
import pednstream_amd.engine as eng
names = ["entry->SlotRec", "SlotRec->batch loaded", "send_flow", "recv_flow", "tf row (dyn)", "LDS write P*s", "barrier 1", "column pass", "barrier 2", "row sums+stores"]
for network in sys.argv[1:]:
    gen = NetworkEnvGenerator(os.path.join(ROOT, "data"))
    R = 1024
    net = gen.create_network(network, verbose=False, n_replicas=R, rng_seed=0)
    e = net.engine()
    for r in range(R):
        for nid in net.origin_nodes:
            e.set_demand(net.nodes[nid].index, replica_demand(net.simulation_steps, r), replica=r)
    net._dirty_demand = set()
    lib = ctypes.CDLL(os.environ["PEDN_HIP_LIB"])
    e.run(1, 150); e.synchronize()
    lib.pedn_debug_phases(None, 1)
    e.run(150, 250); e.synchronize()
    out = (ctypes.c_ulonglong * 16)()
    lib.pedn_debug_phases(out, 0)
    o = np.array(out[:], dtype=np.float64)
    n = o[11]
    print(f"== {network}: {int(n)} active waves over 100 steps; mean wave lifetime {o[12]/n/100:.2f} us (s_memtime at 100 MHz)")
    for i in range(1, 11):
        print(f"   {names[i-1]:26s} {o[i]/n/100:7.3f} us  {100*o[i]/o[12]:5.1f} %")
    net.close()
