"""Vectorised RL environment around the hot path (BASELINE config #5).

Host mirror of the reference's multi-agent glue, with the per-env Python loops replaced by device kernels that act on
every replica at once:

  AgentManager     agent discovery from the controller configuration            rl/discovery.py:20-178
  VecPedNetEnv     n_envs independent copies of one scenario; step(actions[B, A]) ->
                   (obs[B, O] float32, rewards[B, n_agents] float32, terminated, truncated, info)
                   = ActionApplier (rl/builders.py:241-352) -> network_loading x action_gap ->
                     ObservationBuilder (rl/builders.py:25-238) + reward (rl/pz_pednet_env.py:548-581)
  PedNetParallelEnv  single-env facade with the reference's dict-of-agents reset()/step() signature
                   (rl/pz_pednet_env.py:143-254); it does not import pettingzoo/gymnasium.

Layout of one action row: agents in the reference's order (separators first, rl/discovery.py:121-123), one width per
separator, one width per controlled outgoing link of a gater.  Observation row: 4 features per separator,
features_per_link x outdegree per gater (no padding).  ``reward_mode="reference"`` reproduces what the reference
actually computes -- its ``return rewards`` sits inside the agent loop (pz_pednet_env.py:581), so only the first agent
is ever rewarded; ``reward_mode="all"`` rewards every gater.
"""
import numpy as np

OBS_MODES = {"option1": 1, "option2": 2, "option3": 3, "option4": 4, "option5": 5}
FEATURES_PER_LINK = {"option1": 3, "option2": 4, "option3": 5, "option4": 2, "option5": 7}


class Box:
    """Minimal stand-in for ``gymnasium.spaces.Box`` (low / high / shape / dtype / sample / contains); the real class is
    returned instead when gymnasium is installed (rl/spaces.py:41-105 builds the same bounds)."""

    def __init__(self, low, high, shape, dtype=np.float32):
        self.low = np.broadcast_to(np.asarray(low, dtype=dtype), shape).copy()
        self.high = np.broadcast_to(np.asarray(high, dtype=dtype), shape).copy()
        self.shape, self.dtype = tuple(shape), np.dtype(dtype)

    def sample(self, rng=None):
        rng = rng or np.random.default_rng()
        lo = np.where(np.isfinite(self.low), self.low, -1e6)
        hi = np.where(np.isfinite(self.high), self.high, 1e6)
        return rng.uniform(lo, hi).astype(self.dtype)

    def contains(self, x):
        x = np.asarray(x)
        return x.shape == self.shape and bool(np.all(x >= self.low) and np.all(x <= self.high))


def _make_box(low, high, shape):
    try:
        from gymnasium import spaces

        return spaces.Box(low=np.broadcast_to(np.asarray(low, dtype=np.float32), shape).copy(),
                          high=np.broadcast_to(np.asarray(high, dtype=np.float32), shape).copy(), shape=shape, dtype=np.float32)
    except Exception:
        return Box(low, high, shape)


class AgentManager:
    """Maps the scenario's ``controllers`` section to agents (rl/discovery.py)."""

    def __init__(self, network):
        self.network = network
        self.separator_agents, self.gater_agents, self.agent_to_type = {}, {}, {}
        for pair in network.controller_links:
            a, b = sorted(int(x) for x in pair.split("-"))
            fwd, rev = network.links.get((a, b)), network.links.get((b, a))
            if not fwd or not rev:
                raise ValueError(f"Missing bidirectional links for separator {(a, b)}")
            if not fwd.is_separator:
                raise ValueError(f"Link {a}->{b} is not a Separator. Use Separator links for lane control.")
            aid = f"sep_{a}_{b}"
            self.separator_agents[aid] = {"forward": fwd, "reverse": rev, "total_width": fwd._width}
            self.agent_to_type[aid] = "sep"
        for nid in network.controller_gaters:
            if nid not in network.nodes:
                raise ValueError(f"Gater node {nid} not found in network")
            node = network.nodes[nid]
            outs = [l for l in node.outgoing_links if not l.is_virtual and not l.is_separator]
            if not outs:
                raise ValueError(f"Gater node {nid} has no real outgoing links to control")
            aid = f"gate_{nid}"
            self.gater_agents[aid] = {"node": node, "out_links": outs}
            self.agent_to_type[aid] = "gate"
        self.max_outdegree = max((len(g["out_links"]) for g in self.gater_agents.values()), default=0)

    def get_all_agent_ids(self):
        return list(self.separator_agents.keys()) + list(self.gater_agents.keys())

    def get_agent_type(self, agent_id):
        if agent_id not in self.agent_to_type:
            raise ValueError(f"Unknown agent ID: {agent_id}")
        return self.agent_to_type[agent_id]

    def get_separator_links(self, agent_id):
        d = self.separator_agents[agent_id]
        return d["forward"], d["reverse"]

    def get_gater_node(self, agent_id):
        return self.gater_agents[agent_id]["node"]

    def get_gater_outgoing_links(self, agent_id):
        return self.gater_agents[agent_id]["out_links"]

    def get_max_outdegree(self, agent_id):
        return len(self.gater_agents[agent_id]["out_links"])

    def get_gater_action_mask(self, agent_id):
        mask = np.zeros(self.max_outdegree, dtype=np.float32)
        mask[:len(self.gater_agents[agent_id]["out_links"])] = 1.0
        return mask

    def flat_spec(self):
        """(agent ids, type[], link_ptr[], links[]) in the layout of ``pedn_rl_desc``."""
        ids = self.get_all_agent_ids()
        types, ptr, links = [], [0], []
        for aid in ids:
            if self.agent_to_type[aid] == "sep":
                f, r = self.get_separator_links(aid)
                types.append(0)
                links += [f.index, r.index]
            else:
                types.append(1)
                links += [l.index for l in self.get_gater_outgoing_links(aid)]
            ptr.append(len(links))
        return ids, np.array(types, np.int32), np.array(ptr, np.int32), np.array(links, np.int32)


class _DeviceBuffer:
    """A raw device pointer dressed up for ``torch.as_tensor`` (``__cuda_array_interface__``, version 2)."""

    def __init__(self, ptr, shape, typestr):
        self.__cuda_array_interface__ = {"data": (int(ptr), False), "shape": tuple(shape), "typestr": typestr, "version": 2,
                                         "strides": None}


class VecPedNetEnv:
    """``n_envs`` replicas of one scenario stepped together on one GPU."""

    def __init__(self, dataset, n_envs=1, obs_mode="option1", normalize_obs=False, action_gap=1, seed=0,
                 reward_mode="reference", data_dir="data", replica_offset=0, device=0, network=None, verbose=False,
                 history="full"):
        """``history="recent"``: the device keeps only what the recurrence and the observations need (Network's ``history``):
        45_intersections x 2048 envs x T = 700 takes 3.9 GB instead of 19.3 GB; observations, rewards and every number of the
        simulation are the same."""
        from .env_loader import NetworkEnvGenerator

        if obs_mode not in OBS_MODES:
            raise ValueError(f"obs_mode must be one of {list(OBS_MODES)}, got: {obs_mode}")
        self.dataset, self.n_envs, self.obs_mode, self.normalize_obs = dataset, int(n_envs), obs_mode, bool(normalize_obs)
        self.action_gap = int(action_gap)
        self.env_generator = NetworkEnvGenerator(data_dir)
        self.network = network or self.env_generator.create_network(dataset, verbose=verbose, n_replicas=self.n_envs,
                                                                    rng_seed=seed, replica_offset=replica_offset, device=device,
                                                                    history=history)
        self.simulation_steps = self.network.params["simulation_steps"]
        self.scenarios = None          # the ScenarioBatch of the last randomised reset
        self._ext_stream = None        # torch view of the engine's stream (step_device(sync=False))
        self._init_widths = None       # [L, 3] initial front / back / separator widths (reset)
        ut = self.network.params["unit_time"]
        self._max_delta_sep_width = 0.25 * ut          # pz_pednet_env.py:84-86
        self._max_delta_gate_width = 0.25 * ut
        self._min_sep_width = 1.5
        self.agent_manager = AgentManager(self.network)
        self.possible_agents, types, ptr, links = self.agent_manager.flat_spec()
        if not self.possible_agents:
            raise ValueError("scenario defines no controller agents")
        self.features_per_link = FEATURES_PER_LINK[obs_mode]
        eng = self.network.engine()
        self.n_actions, self.n_obs = eng.rl_configure(types, ptr, links, OBS_MODES[obs_mode], normalize_obs,
                                                      {"reference": 0, "all": 1}[reward_mode], self._max_delta_sep_width,
                                                      self._max_delta_gate_width, self._min_sep_width)
        self._types, self._ptr = types, ptr
        # per-agent slices of the flat rows and action bounds (rl/spaces.py:41-73)
        self.action_slices, self.obs_slices, lo, hi = {}, {}, [], []
        a0 = o0 = 0
        for aid, ty, p0, p1 in zip(self.possible_agents, types, ptr[:-1], ptr[1:]):
            if ty == 0:
                w = self.agent_manager.separator_agents[aid]["total_width"]
                na, no = 1, 4
                lo.append(self._min_sep_width)
                hi.append(w - self._min_sep_width)
            else:
                outs = self.agent_manager.get_gater_outgoing_links(aid)
                na, no = len(outs), len(outs) * self.features_per_link
                lo += [0.0] * na
                hi += [l.width for l in outs]
            self.action_slices[aid] = slice(a0, a0 + na)
            self.obs_slices[aid] = slice(o0, o0 + no)
            a0 += na
            o0 += no
        self.action_low, self.action_high = np.array(lo, np.float32), np.array(hi, np.float32)
        self.sim_step = 1
        self._device_views = None

    # ------------------------------------------------------------------------------------------------ API
    AUTO_VECTORISED_FROM = 65      # randomize(mode="auto"): batches of this many envs and more draw their scenarios on the device

    def randomize(self, seed=None, mode="auto"):
        """Draw a new scenario for every env and upload it per replica (``pednstream_amd.scenarios``).

        ``mode="reference"``: the reference's randomisers, one np.random stream consumed env after env
        (env_loader.py:183-259,363-424: link parameters of 20 % of the corridors, OD weights, demand pattern / lambdas) --
        bit-exact against the reference (goldens ``rand_*``) and host-bound, ~0.2 ms per env: 0.4 s for 2048 envs, twenty times an
        episode's stepping time, which no prefetch thread can hide (it is ONE numpy stream, and the stepping loop holds the GIL).
        ``mode="auto"`` (the default): "reference" up to 64 envs -- a drop-in single env or a small batch reproduces the reference's
        numbers --, "vectorised" from 65 on.  ``mode="vectorised"``: the same distributions drawn for all envs at once ON THE DEVICE
        (``ScenarioBatch.draw_random`` -> ``pedn_randomize_scenarios``: nothing is uploaded); not the reference's random numbers.  Two deliberate differences from
        ``randomize_network`` in both modes: every env is perturbed from the BASE configuration (the reference keeps
        perturbing its already perturbed config), and ``generate_random_od_nodes`` is not applied because it changes the
        topology."""
        from .scenarios import ScenarioBatch

        gen = self.env_generator
        if gen.config is None or gen.network_data is None:
            gen.network_data = gen.load_network_data(self.dataset)
        if mode == "auto":
            mode = "vectorised" if self.n_envs >= self.AUTO_VECTORISED_FROM else "reference"
        if mode == "vectorised" and self.scenarios is not None:
            batch = self.scenarios                 # nothing host-side to rebuild: the device draws in place (0.5 ms of Python per reset saved)
        else:
            batch = ScenarioBatch(self.network, edge_distances=gen.network_data["edge_distances"])
        if mode == "vectorised":
            batch.draw_random(seed)
        elif mode == "vectorised_host":           # the numpy generator the device kernels are cross-checked against
            batch.draw_random_host(seed)
        elif mode == "reference":
            if seed is not None:
                np.random.seed(seed)
            for r in range(self.n_envs):
                batch.set_replica(r, link_params_overrides=gen.generate_random_link_params(None),
                                  od_flows=gen.generate_random_od_flows(None) if self.network.od_manager is not None else None,
                                  demand_params_overrides=gen.generate_random_demand_params(None))
        else:
            raise ValueError(f"unknown randomisation mode {mode!r}")
        batch.commit(reset=False)
        self.scenarios = batch

    def reset(self, options=None, seed=None):
        """All replicas back to t = 0: histories cleared, widths back to their initial values.  ``options={'randomize':
        True}`` additionally draws a new scenario per env (pz_pednet_env.py:143-193 rebuilds the network instead);
        ``'mode'``: ``'reference'`` (the reference's np.random stream, env after env), ``'vectorised'`` (drawn on the device) or
        ``'auto'`` (default: reference up to 64 envs, vectorised above -- see ``randomize``)."""
        net = self.network
        eng = net.engine()
        if options and options.get("randomize", False):
            self.randomize(seed, mode=options.get("mode", "auto"))
        net.reset(lazy=True)
        # every env starts from the same widths: broadcast on the device instead of four [L, R] uploads; the host mirrors (four [L, R]
        # arrays, 11 MB at 2048 envs) are not rebuilt -- they are marked stale and come back from the device if somebody looks
        if self._init_widths is None:
            self._init_widths = np.array([l._init_widths for l in net._link_list], dtype=np.float64).reshape(net.n_links, 3)
        init = self._init_widths
        eng.reset_widths(init[:, 0].copy(), init[:, 1].copy(), init[:, 2].copy())
        net._widths_stale = True
        net._tf_host = {}
        self.sim_step = 1
        obs, _ = eng.rl_observe(self.sim_step, accumulate=False)
        return obs, {}

    def step(self, actions, fetch=True):
        """actions [n_envs, n_actions] widths in metres (None = keep the current widths)."""
        if self.sim_step + self.action_gap - 1 > self.simulation_steps:
            raise IndexError("episode is over; call reset()")
        eng = self.network._flush()
        obs, rew = eng.rl_step(actions, self.sim_step, self.action_gap, fetch=fetch)
        self.sim_step += self.action_gap
        self.network.current_step = self.sim_step - 1
        self.network._widths_stale = True
        terminated = (self.sim_step - 1) >= self.simulation_steps     # pz_pednet_env.py:592 evaluated before the increment
        return obs, rew, terminated, False, {}

    def step_async(self, actions):
        """``step`` in two halves for a caller that steps several batched envs before it looks at any of them (``MultiScenarioVecEnv``):
        the launches are enqueued, nothing is waited for; ``step_wait`` fetches."""
        if self.sim_step + self.action_gap - 1 > self.simulation_steps:
            raise IndexError("episode is over; call reset()")
        self.network._flush().rl_step(actions, self.sim_step, self.action_gap, fetch=False)
        self.sim_step += self.action_gap
        self.network.current_step = self.sim_step - 1
        self.network._widths_stale = True

    def step_wait(self):
        obs, rew = self.network.engine().rl_fetch()
        return obs, rew, (self.sim_step - 1) >= self.simulation_steps, False, {}

    def step_device(self, actions, sync=True):
        """``step`` for an on-GPU learner: ``actions`` is a torch CUDA tensor (float64, [n_envs, n_actions], contiguous) or None;
        returns ``(obs, rewards, terminated)`` where obs [n_envs, n_obs] and rewards [n_envs, n_agents] are float32 torch tensors
        that ALIAS the engine's device buffers (``pedn_rl_device_ptr``) -- no host copy.  They are overwritten by the next
        step, so clone what must be kept.  The call waits for the caller's current torch stream before launching and for the
        engine's stream before returning, so plain sequential use is safe.

        ``sync=False``: no host synchronisation at all -- the engine's stream waits (on the device) for the caller's current torch
        stream, and that stream then waits for the engine's: the policy's kernels, the env step and whatever consumes the observations
        are chained by events, and the host runs ahead enqueueing the next step (45_intersections x 2048 envs end to end:
        70.7 -> 66.4 us per step with a random torch policy, 172 -> 163 with a small MLP: the loop is bound by torch's own per-op
        launch overhead -- ``capture`` below takes the host out of it: 40 / 61 us, profiles/r05_graph_rollout.txt).  Work issued on
        OTHER torch streams must be ordered by the caller.

        torch and the engine share one HIP runtime whichever is imported first (``engine._bind_hip_runtime``)."""
        import torch

        if self.sim_step + self.action_gap - 1 > self.simulation_steps:
            raise IndexError("episode is over; call reset()")
        eng = self.network._flush()
        ptr = 0
        if actions is not None:
            if not (actions.is_cuda and actions.dtype == torch.float64 and actions.is_contiguous()
                    and tuple(actions.shape) == (self.n_envs, self.n_actions)):
                raise ValueError(f"actions must be a contiguous float64 CUDA tensor of shape {(self.n_envs, self.n_actions)}")
            ptr = actions.data_ptr()
        if not sync:
            dev = torch.device("cuda", self.network.device)
            if self._ext_stream is None:
                self._ext_stream = torch.cuda.ExternalStream(eng.stream_ptr(), device=dev)
            cur = torch.cuda.current_stream(dev)
            self._ext_stream.wait_stream(cur)                       # the actions are ready when the engine's launches start
            if ptr:
                eng.rl_step_device(ptr, self.sim_step, self.action_gap, ordered=True)
            else:
                eng.rl_step(None, self.sim_step, self.action_gap, fetch=False, ordered=True)
            cur.wait_stream(self._ext_stream)                       # whatever the caller enqueues next sees this step's results
        else:
            if actions is not None:
                torch.cuda.current_stream(actions.device).synchronize()
            if ptr:
                eng.rl_step_device(ptr, self.sim_step, self.action_gap)
            else:
                eng.rl_step(None, self.sim_step, self.action_gap, fetch=False)
            eng.synchronize()
        self.sim_step += self.action_gap
        self.network.current_step = self.sim_step - 1
        self.network._widths_stale = True
        obs, rew = self.device_views()
        return obs, rew, (self.sim_step - 1) >= self.simulation_steps

    def device_views(self):
        """(obs [n_envs, n_obs], rewards [n_envs, n_agents]): float32 torch tensors that ALIAS the engine's device buffers."""
        if self._device_views is None:
            import torch

            eng = self.network.engine()
            dev = torch.device("cuda", self.network.device)
            view = lambda which, cols: torch.as_tensor(_DeviceBuffer(eng.rl_device_ptr(which), (self.n_envs, cols), "<f4"), device=dev)
            self._device_views = (view(1, self.n_obs), view(2, len(self.possible_agents)))
        return self._device_views

    def capture(self, policy_fn, on_step=None, generators=(), steps_per_replay=1):
        """A graph-replayable rollout loop: ``policy_fn(obs) -> actions`` (torch ops only; obs is the engine's float32 observation buffer
        [n_envs, n_obs], actions a contiguous float64 CUDA tensor [n_envs, n_actions]) followed by one env step and ``on_step(obs, rewards)``
        (optional, torch ops only: reward bookkeeping, storing the transition) captured ONCE as a ``torch.cuda.CUDAGraph`` and replayed
        per policy step -- see ``GraphedRollout``.  ``generators``: every ``torch.Generator`` the two callables draw from other than the
        default one (torch must know them before the capture: ``CUDAGraph.register_generator_state``).  ``steps_per_replay`` > 1: that
        many consecutive iterations in ONE graph (``GraphedRollout.step`` then advances by that many policy steps; the tail of an episode
        that does not fill a graph is stepped eagerly)."""
        return GraphedRollout(self, policy_fn, on_step, generators, steps_per_replay)

    def _ordered_behind_engine(self):
        """The caller's current torch stream waits (on the device) for everything enqueued on the engine's stream so far."""
        import torch

        dev = torch.device("cuda", self.network.device)
        if self._ext_stream is None:
            self._ext_stream = torch.cuda.ExternalStream(self.network.engine().stream_ptr(), device=dev)
        torch.cuda.current_stream(dev).wait_stream(self._ext_stream)

    def gather_device(self, total_envs=None):
        """Observations and rewards of ALL ranks' envs after ``step_device``: (obs [n_envs_total, n_obs], rewards
        [n_envs_total, n_agents]) float32 torch tensors on this rank's GPU, rows ordered by global env id -- the engine's
        buffers go into the collective as they are (RCCL all_gather with backend "nccl"; no host copy)."""
        from .ensemble import gather_replica_summaries

        if self._device_views is None:
            raise RuntimeError("call step_device() first")
        obs, rew = self._device_views
        return gather_replica_summaries(obs, total_envs), gather_replica_summaries(rew, total_envs)

    def split_obs(self, obs_row):
        return {aid: obs_row[..., sl] for aid, sl in self.obs_slices.items()}

    def close(self):
        self._closed = True          # a GraphedRollout of this env must not replay launches that carry the freed engine's pointers
        self.network.close()


class GraphedRollout:
    """Config #5 the way BASELINE words it -- the env FEEDING a policy (rl/pz_pednet_env.py:195-254 is called once per policy step,
    rl/train_ppo_sb3.py:246) -- without the host in the loop: one iteration = policy forward -> env step -> ``on_step`` is captured
    once as a ``torch.cuda.CUDAGraph`` and replayed.  What makes the env step replayable is the engine's device-resident step clock
    (include/pedn.h: pedn_rl_clock_begin / pedn_rl_step_clocked): the step's launches have constant arguments, the step index is read
    from -- and advanced in -- device memory.

        roll = env.capture(policy_fn, on_step)
        env.reset()
        while not roll.step(): pass          # one policy step per call; True when the episode is over

    The first step of an episode runs eagerly (its stand-alone turning fractions read the gate widths behind that step's actions, which
    only the ordinary step orders); so does a step after anything else touched the engine (``env.step``, a setter, a read): the rollout
    notices (``pedn_rl_clocked``) and begins a new clocked section.  The same numbers as ``step_device`` step for step
    (tests/test_gpu_rl.py)."""

    def __init__(self, env, policy_fn, on_step=None, generators=(), steps_per_replay=1):
        import torch

        self.env, self.policy_fn, self.on_step, self.generators = env, policy_fn, on_step, tuple(generators)
        self.n = max(1, int(steps_per_replay))
        self.dev = torch.device("cuda", env.network.device)
        self.obs, self.rew = env.device_views()
        self.graph = None
        self._actions = None           # the policy's output tensor of the captured iteration (its address is what the graph reads)
        self.replays = self.eager_steps = self.recaptures = 0
        self._signature, self._verify, self._in_flight = None, True, False

    def _iteration(self, stream_ptr):
        a = self.policy_fn(self.obs)
        env = self.env
        if not (a.is_cuda and a.dtype.is_floating_point and a.element_size() == 8 and a.is_contiguous() and tuple(a.shape) == (env.n_envs, env.n_actions)):
            raise ValueError(f"policy_fn must return a contiguous float64 CUDA tensor of shape {(env.n_envs, env.n_actions)}")
        env.network.engine().rl_step_clocked(a.data_ptr(), env.action_gap, stream_ptr)
        if self.on_step is not None:
            self.on_step(self.obs, self.rew)
        return a

    def step(self):
        """One policy step for every env -- ``steps_per_replay`` of them when a whole graph fits what is left of the episode; returns
        True when the episode is over (``env.reset()`` next)."""
        import torch

        env = self.env
        if getattr(env, "_closed", False):
            raise RuntimeError("the environment of this rollout is closed")
        if env.sim_step + env.action_gap - 1 > env.simulation_steps:
            raise IndexError("episode is over; call reset()")
        eng = env.network._flush() if not env.network.engine().rl_clocked() else env.network.engine()
        fits = env.sim_step + self.n * env.action_gap - 1 <= env.simulation_steps
        with torch.cuda.device(self.dev):
            if not eng.rl_clocked() or not fits:
                if self._in_flight:
                    # Replays are still running.  The eager step below ends the clocked section, which waits for the device anyway -- but
                    # it would first enqueue the policy's kernels and a cross-stream wait behind the replays, and a second hardware queue
                    # with a blocked wait slows every dispatch of the busy one (measured: 174 replays 44.7 -> 51.5 ms).  Wait first.
                    torch.cuda.current_stream(self.dev).synchronize()
                    self._in_flight = False
                self._verify = True
                begun = False
                if env.sim_step > 1 and fits:
                    try:
                        eng.rl_clock_begin(env.sim_step)
                        begun = True
                    except RuntimeError:
                        begun = False          # the fractions of this step are not prepared (a setter came in between): one eager step
                if not begun:
                    a = self.policy_fn(self.obs)
                    _, _, done = env.step_device(a, sync=False)
                    if self.on_step is not None:
                        self.on_step(self.obs, self.rew)
                    self.eager_steps += 1
                    if not done and env.sim_step + self.n * env.action_gap - 1 <= env.simulation_steps:
                        eng.rl_clock_begin(env.sim_step)
                        env._ordered_behind_engine()
                    return done
                env._ordered_behind_engine()
            if self._verify:
                # the captured launches carry the engine's device view by value: a scenario switch since the capture (the first
                # randomised reset brings the per-replica-parameter kernels) makes the graph stale -- captured again.  Only something
                # that ended the clocked section can have changed it, so the check runs once per section.
                self._verify = False
                if self.graph is not None and eng.rl_clock_signature() != self._signature:
                    self.graph = None
                    self.recaptures += 1
            if self.graph is None:
                self._capture()
            self.graph.replay()
            self._in_flight = True
        self.replays += 1
        env.sim_step += self.n * env.action_gap
        env.network.current_step = env.sim_step - 1
        env.network._widths_stale = True
        return (env.sim_step - 1) >= env.simulation_steps

    def _capture(self):
        import torch

        # (no warm-up calls of policy_fn here: they would advance whatever state the policy keeps.  The eager step every episode starts
        # with has already run it once, so what torch sets up lazily exists.)
        side = torch.cuda.Stream(self.dev)
        side.wait_stream(torch.cuda.current_stream(self.dev))
        g = torch.cuda.CUDAGraph()
        for gen in self.generators:
            g.register_generator_state(gen)
        with torch.cuda.graph(g, stream=side):  # recorded, not run: the env does not advance
            self._actions = [self._iteration(torch.cuda.current_stream(self.dev).cuda_stream) for _ in range(self.n)]
        self.graph = g
        self._signature = self.env.network.engine().rl_clock_signature()


class PedNetParallelEnv:
    """Single-env facade with the reference's dict-of-agents interface (rl/pz_pednet_env.py:37-254)."""

    metadata = {"render_modes": [], "name": "pednet_v0"}

    def __init__(self, dataset, normalize_obs=False, obs_mode="option1", render_mode=None, verbose=False, action_gap=1,
                 seed=None, **kw):
        self._vec = VecPedNetEnv(dataset, n_envs=1, obs_mode=obs_mode, normalize_obs=normalize_obs, action_gap=action_gap,
                                 seed=0 if seed is None else seed, verbose=verbose, **kw)
        self.network = self._vec.network
        self.agent_manager = self._vec.agent_manager
        self.possible_agents = list(self._vec.possible_agents)
        self.simulation_steps = self._vec.simulation_steps
        self._cumulative_rewards = {a: 0.0 for a in self.possible_agents}
        self._action_spaces, self._observation_spaces = {}, {}
        self.render_mode = render_mode
        # what the reference's trainers and rule-based agents read off the env (rl/rl_utils.py:133,197: env.obs_builder.features_per_link;
        # rl/agents/rule_based.py:189, rl/train_rl.py:177: env.obs_mode)
        self.obs_mode, self.normalize_obs, self.action_gap = obs_mode, bool(normalize_obs), int(action_gap)
        self.obs_builder = type("ObservationBuilderView", (), {"features_per_link": self._vec.features_per_link, "obs_mode": obs_mode,
                                                               "normalize": bool(normalize_obs)})()
        self.dataset = dataset

    @property
    def agents(self):
        return self.possible_agents.copy()

    @property
    def sim_step(self):
        return self._vec.sim_step

    def action_space(self, agent):
        """Physical width bounds (rl/spaces.py:41-73): [min_sep, width - min_sep] for a separator, [0, link width] per
        controlled link for a gater."""
        if agent not in self._vec.action_slices:
            raise ValueError(f"Agent {agent} not found in action spaces")
        if agent not in self._action_spaces:      # one object per agent (rl/pz_pednet_env.py:136 caches; PettingZoo's API test asserts identity)
            sl = self._vec.action_slices[agent]
            self._action_spaces[agent] = _make_box(self._vec.action_low[sl], self._vec.action_high[sl], (sl.stop - sl.start,))
        return self._action_spaces[agent]

    def observation_space(self, agent):
        if agent not in self._vec.obs_slices:
            raise ValueError(f"Agent {agent} not found in observation spaces")
        if agent not in self._observation_spaces:
            sl = self._vec.obs_slices[agent]
            self._observation_spaces[agent] = _make_box(-np.inf, np.inf, (sl.stop - sl.start,))
        return self._observation_spaces[agent]

    def _dict(self, row):
        return {a: np.array(row[sl]) for a, sl in self._vec.obs_slices.items()}

    def _infos(self):
        """rl/pz_pednet_env.py:631-642"""
        return {a: {"step": self.sim_step, "cumulative_reward": self._cumulative_rewards.get(a, 0.0)} for a in self.possible_agents}

    def seed(self, seed):
        """rl/pz_pednet_env.py:118-122: seeds the global generators the host-side randomisers draw from."""
        import random

        self._seed = seed
        np.random.seed(seed)
        random.seed(seed)

    def reset(self, seed=None, options=None):
        """``seed`` is ignored like in the reference (:143-193; it seeds at construction); ``options={'randomize': True}`` draws
        a new scenario (link parameters, OD weights, demand -- not the OD nodes, see ``VecPedNetEnv.randomize``) from the
        global np.random stream instead of rebuilding the network."""
        obs, _ = self._vec.reset(options=options, seed=None)
        self._cumulative_rewards = {a: 0.0 for a in self.possible_agents}
        return self._dict(obs[0]), self._infos()

    def step(self, actions):
        for a in actions:
            if a not in self.possible_agents:
                raise ValueError(f"Unknown agent: {a}")
        row = None
        if len(actions) > 0:
            # agents without an action are left alone (apply_all_actions only touches the given agents): NaN = skip the slot
            row = np.full((1, self._vec.n_actions), np.nan)
            for aid, sl in self._vec.action_slices.items():
                if aid in actions:
                    row[0, sl] = np.asarray(actions[aid], dtype=np.float64).reshape(-1)
        obs, rew, term, trunc, _ = self._vec.step(row)
        rewards = {a: float(rew[0, i]) for i, a in enumerate(self.possible_agents)}
        for a, r in rewards.items():
            self._cumulative_rewards[a] += r
        return (self._dict(obs[0]), rewards, {a: term for a in self.possible_agents},
                {a: False for a in self.possible_agents}, self._infos())

    def save(self, simulation_dir, base_dir="../outputs"):
        """rl/pz_pednet_env.py:688-691: the reference's JSON files, written from the device histories."""
        from .output_handler import OutputHandler

        OutputHandler(base_dir=base_dir, simulation_dir=simulation_dir).save_network_state(self.network)

    def render(self, *a, **k):
        """Plotting (NetworkVisualizer) is outside the hot path: nothing to do without a render mode, like the reference; with one
        (rl/rl_example.py passes render_mode="animate" and calls render() last) it warns and returns -- save() has written the files
        the reference's visualiser reads."""
        if self.render_mode is None:
            return None
        import warnings

        warnings.warn("PedNetParallelEnv.render: rendering is not provided by pednstream_amd; save() writes the files the reference's "
                      "NetworkVisualizer reads", stacklevel=2)
        return None

    def close(self):
        self._vec.close()


class MultiScenarioVecEnv:
    """Vectorised env whose groups of ``group_size`` envs each live in their OWN randomised scenario, topology included.

    ``NetworkEnvGenerator.randomize_network`` (src/utils/env_loader.py:160-181) also moves origin / destination nodes,
    which changes which nodes own virtual links and therefore the node classes and routes: such scenarios cannot share
    one engine.  Here every group is a separate ``Network`` + engine (own HIP stream, same GPU) built by the mirrored
    ``randomize_network``; inside a group the envs share the scenario and differ by RNG key.  Agents are defined by the
    scenario's ``controllers`` section, which the randomiser never touches, so action / observation layouts agree."""

    def __init__(self, dataset, n_envs, group_size=64, obs_mode="option1", normalize_obs=False, action_gap=1, seed=0,
                 reward_mode="reference", data_dir="data", device=0):
        from .env_loader import NetworkEnvGenerator

        self.dataset, self.n_envs, self.group_size = dataset, int(n_envs), int(group_size)
        self.kw = dict(obs_mode=obs_mode, normalize_obs=normalize_obs, action_gap=action_gap, reward_mode=reward_mode,
                       data_dir=data_dir)
        self.seed, self.device, self.data_dir = int(seed), device, data_dir
        self.sizes = [min(self.group_size, self.n_envs - o) for o in range(0, self.n_envs, self.group_size)]
        self.generators = [NetworkEnvGenerator(data_dir) for _ in self.sizes]
        self.groups = []
        off = 0
        for gen, size in zip(self.generators, self.sizes):
            net = gen.create_network(dataset, verbose=False, n_replicas=size, replica_offset=off, rng_seed=self.seed, device=device)
            self.groups.append(VecPedNetEnv(dataset, n_envs=size, network=net, **self.kw))
            off += size
        g0 = self.groups[0]
        self.possible_agents, self.n_actions, self.n_obs = g0.possible_agents, g0.n_actions, g0.n_obs
        self.action_low, self.action_high = g0.action_low, g0.action_high
        self.simulation_steps = g0.simulation_steps

    def reset(self, options=None, seed=None):
        randomize = bool(options and options.get("randomize", False))
        obs = []
        off = 0
        for k, (gen, size) in enumerate(zip(self.generators, self.sizes)):
            if randomize:
                self.groups[k].close()
                net = gen.randomize_network(self.dataset, seed=None if seed is None else seed + k, verbose=False, n_replicas=size,
                                            replica_offset=off, rng_seed=self.seed, device=self.device)
                self.groups[k] = VecPedNetEnv(self.dataset, n_envs=size, network=net, **self.kw)
                if self.groups[k].possible_agents != self.possible_agents:
                    raise RuntimeError("randomisation changed the agent set")
            o, _ = self.groups[k].reset()
            obs.append(o)
            off += size
        return np.concatenate(obs, axis=0), {}

    def step(self, actions):
        from .engine import Engine

        actions = np.asarray(actions, dtype=np.float64)
        g0 = self.groups[0]
        if g0.sim_step + g0.action_gap - 1 > g0.simulation_steps:
            raise IndexError("episode is over; call reset()")
        # every group's launches first (own engine, own stream: they overlap), then one fetch each -- in ONE library call
        # (pedn_rl_step_many; group by group from Python: 33 us per engine and step; from 2-16 host threads: slower still)
        obs, rew = Engine.rl_step_many([g.network._flush() for g in self.groups], actions, g0.sim_step, g0.action_gap)
        for g in self.groups:
            g.sim_step += g.action_gap
            g.network.current_step = g.sim_step - 1
            g.network._widths_stale = True
        return obs, rew, (g0.sim_step - 1) >= g0.simulation_steps, False, {}

    def close(self):
        for g in self.groups:
            g.close()
