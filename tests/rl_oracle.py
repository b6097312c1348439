"""CPU restatement of the reference's RL glue on top of the C oracle -- TEST INFRASTRUCTURE.

ActionApplier (rl/builders.py:281-352), ObservationBuilder (:68-238) and PedNetParallelEnv._compute_rewards
(rl/pz_pednet_env.py:548-581) written with numpy scalars so that every float32/float64 promotion is numpy's own.
Pinned against the goldens rl_*.npz captured from the real reference modules (tests/test_rl_golden.py)."""
import numpy as np

import oracle_driver as od

FPL = {"option1": 3, "option2": 4, "option3": 5, "option4": 2, "option5": 7}


class RlOracle:
    def __init__(self, net, model, spec, obs_mode, normalize, action_gap, seed=0, replica=0, reward_mode="reference",
                 link_kc=None, link_kj=None):
        """spec: list of {"id", "type", "links": [link ids "u_v"]} in agent order.  link_kc / link_kj: k_critical / k_jam per
        link of THIS replica when the batch carries per-replica scenarios (the reward and option4 read them, pz_pednet_env.py:569,
        builders.py:157); default: the network's own."""
        self.net, self.model = net, model
        self.kc = None if link_kc is None else np.asarray(link_kc, dtype=np.float64)
        self.kj = None if link_kj is None else np.asarray(link_kj, dtype=np.float64)
        self.o = od.Oracle(model, seed=seed, replica=replica)
        self.obs_mode, self.normalize, self.gap, self.reward_mode = obs_mode, normalize, action_gap, reward_mode
        self.links = list(net.links.values())
        by_id = {l.link_id: l for l in self.links}
        self.agents = [(a["id"], a["type"], [by_id[x] for x in a["links"]]) for a in spec]
        ut = net.params["unit_time"]
        self.max_delta_sep = self.max_delta_gate = 0.25 * ut
        self.min_sep = 1.5
        L = model["n_links"]
        self.front = np.array([l.front_gate_width for l in self.links], dtype=np.float64)
        self.back = np.array([l.back_gate_width for l in self.links], dtype=np.float64)
        self.sep = np.array([l.separator_width if l.is_separator else s0 for l, s0 in zip(self.links, model["sep_width0"])], dtype=np.float64)
        self.sepnp = np.zeros(L)
        self.t = 1
        assert L == len(self.links)

    def _push(self):
        for l in range(len(self.links)):
            self.o.set_width(0, l, self.front[l])
            self.o.set_width(1, l, self.back[l])
            self.o.set_width(2, l, self.sep[l])
            self.o.set_width(3, l, self.sepnp[l])

    def apply(self, row):
        k = 0
        for aid, ty, links in self.agents:
            if ty == "sep":
                f = links[0]
                a = float(np.float32(row[k])); k += 1
                if a != a:          # NaN: the agent is not in the action dict (apply_all_actions only touches the given agents)
                    continue
                cur = self.sep[f.index]
                if abs(a - cur) > self.max_delta_sep:
                    a = cur + np.clip(a - cur, -self.max_delta_sep, self.max_delta_sep)
                a = float(np.clip(a, self.min_sep, f.width - self.min_sep))
                r = f.reverse_link.index
                self.sep[f.index] = self.front[f.index] = self.back[f.index] = a
                self.sep[r] = self.front[r] = self.back[r] = f._width - a
                self.sepnp[f.index] = self.sepnp[r] = 1.0        # np.clip hands the setter an np.float64
            else:
                for l in links:
                    a = float(np.float32(row[k])); k += 1
                    if a != a:
                        continue
                    cur = self.back[l.index]
                    if abs(a - cur) > self.max_delta_gate:
                        a = cur + np.clip(a - cur, -self.max_delta_gate, self.max_delta_gate)
                    a = float(np.clip(a, 0.0, l.width))
                    self.back[l.index] = a
                    self.front[l.reverse_link.index] = a
        self._push()

    def _dens(self, f, l, t):
        n = f["num_pedestrians"]
        if l.is_separator:
            return f["density"][l.index, t]
        return (n[l.index, t] + n[l.reverse_link.index, t]) / np.float32(l.length * l._width)

    def observe(self, t):
        f = {name: self.o.field(name) for name in ("inflow", "outflow", "num_pedestrians", "density", "speed", "travel_time")}
        obs, rewards = [], []
        for ai, (aid, ty, links) in enumerate(self.agents):
            if ty == "sep":
                a, b = links
                x = np.array([f["inflow"][a.index, t], f["outflow"][a.index, t], f["inflow"][b.index, t], f["outflow"][b.index, t]], dtype=np.float32)
                if self.normalize and self.obs_mode in ("option1", "option2"):
                    x = x / np.float32(20.0)
                obs.append(x)
                rewards.append(np.float32(0.0))
                continue
            fpl = FPL[self.obs_mode]
            x = np.zeros(len(links) * fpl, dtype=np.float32)
            lr = 0.0
            dens_all = []
            for i, l in enumerate(links):
                r = l.reverse_link
                dens = self._dens(f, l, t)
                gate = self.back[l.index]
                feats = {"option1": [f["inflow"][l.index, t], f["outflow"][r.index, t], gate],
                         "option2": [f["inflow"][l.index, t], f["outflow"][r.index, t], dens, gate],
                         "option3": [f["inflow"][l.index, t], f["outflow"][l.index, t], f["inflow"][r.index, t], f["outflow"][r.index, t], gate],
                         "option4": [dens / (l.k_jam if self.kj is None else float(self.kj[l.index])), gate],
                         "option5": [f["inflow"][l.index, t], f["outflow"][l.index, t], f["inflow"][r.index, t], f["outflow"][r.index, t],
                                     f["speed"][l.index, t], dens, gate]}[self.obs_mode]
                x[i * fpl:(i + 1) * fpl] = feats
                lr -= f["travel_time"][l.index, t] + f["travel_time"][r.index, t]
                if dens > 4:
                    lr -= 10 * (dens - (l.k_critical if self.kc is None else float(self.kc[l.index])))
                dens_all.append(dens)
            if self.normalize:
                for i in range(len(links)):
                    s = i * fpl
                    if self.obs_mode in ("option1", "option2"):
                        x[s] /= 20.0; x[s + 1] /= 20.0
                    elif self.obs_mode == "option3":
                        x[s] /= 6.0; x[s + 1] /= 20.0; x[s + 2] /= 20.0
            if len(dens_all) > 1:
                avg = np.mean(dens_all)
                lr -= 10.0 * np.mean(np.abs(np.array(dens_all) - avg))
            obs.append(x)
            rewards.append(np.float32(lr))
        if self.reward_mode == "reference":           # `return rewards` inside the loop, pz_pednet_env.py:581
            rewards = [rw if i == 0 else np.float32(0.0) for i, rw in enumerate(rewards)]
        return np.concatenate(obs), np.array(rewards, dtype=np.float32)

    def step(self, row):
        if row is not None:
            self.apply(row)
        cum = None
        for _ in range(self.gap):
            self.o.step(self.t)
            obs, rew = self.observe(self.t)
            cum = rew if cum is None else (cum + rew).astype(np.float32)
            self.t += 1
        return obs, cum
