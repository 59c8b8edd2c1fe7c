"""Oracle of BottleneckAccelEnv with RL vehicles (flow/envs/bottleneck.py:486-757) on top of the open-network oracle.
TEST INFRASTRUCTURE ONLY (tests/ and bench.py's cpu_baseline may import it; flow_amd/ must not).

The simulator step is MergeOracle's (oracle/opennet.py: lane-drop network, head ENV_BOTTLENECK, one acceleration column
per RL slot, NaN = no command).  Around it this module restates, statement by statement and with the reference's own
data structures (per-edge / per-lane lists sorted by position, bisect_left, next_edge / prev_edge walks):

  A1  _apply_rl_actions                       flow/envs/bottleneck.py:662-690
      - actions pair up with the RL vehicles sorted by get_x_by_id; direction = round(.), zeroed while
        time_counter <= lane_change_duration + get_last_lc (THIS FORK's get_last_lc returns the headway,
        flow/core/kernel/vehicle/traci.py:604-614); target lane clipped to the lanes of the vehicle's edge (:978-990)
  A2  additional_command (add_rl_if_exit)     flow/envs/bottleneck.py:733-757
  A3  get_state                               flow/envs/bottleneck.py:539-640
      - per-lane leaders / followers / headways / tailways: flow/core/kernel/vehicle/traci.py:776-950
      - quirks kept: vel_behind of a lane without follower is get_speed('') / max_speed = -1001 / max_speed (:608 tests
        the list, not the entry); the final padding of the relative block ignores the scaling (:616-619)
  A4  compute_reward                          flow/envs/bottleneck.py:642-649, flow/core/rewards.py:6-59, 91-114

SUMO-side statements (third-party, absent: PARITY UNPINNED like every M-rule):
  A5  a commanded lane change is executed with the step's move (SUMO changes lanes after moving); with
      lane_change_mode != 0 it is refused when the vehicle would overlap one on the target lane (ML3 of oracle/refsim.py);
      the vehicle continues on the lowest entry lane of its new lane (M11)
  A6  a re-added RL vehicle enters at the end of a step, front at position 0 of edge "1" on lane
      index % MAX_LANES * scaling with speed min(maxSpeed, speed limit) ("max"), once the insertion test M3 lets it
  A7  the internal lanes netconvert generates at the zipper nodes are one per connection of
      flow/networks/bottleneck.py:179-201 (":4_0" has the lanes of edge "3", lane i leads to lane i // 2 of edge "4")

spec["accel_env"]: path [(edge name, length, lanes)] in driving order (internal edges included), connections
{edge: {fromLane: toLane}}, edge_list [names of get_edge_list()], edge_length {name: m}, rl_names {slot: id},
lane_change_duration, scaling, add_rl_if_exit, max_speed (network), lane_change_mode {slot: mode}.
"""
from bisect import bisect_left

import numpy as np

from .opennet import MergeOracle

MAX_LANES = 4                                   # flow/envs/bottleneck.py:36


class BottleneckAccelOracle(MergeOracle):
    def __init__(self, spec, dtype=np.float64):
        super().__init__(spec, dtype)
        ae = spec["accel_env"]
        self.path = [e for e, _, _ in ae["path"]]
        self.e_len = {e: float(ln) for e, ln, _ in ae["path"]}
        self.e_len.update({k: float(v) for k, v in ae["edge_length"].items()})
        self.e_lanes = {e: int(n) for e, _, n in ae["path"]}
        s, self.e_start = 0.0, {}
        for e, ln, _ in ae["path"]:
            self.e_start[e] = s
            s += float(ln)
        self.conn = {k: {int(a): int(b) for a, b in v.items()} for k, v in ae["connections"].items()}
        self.rl_names = dict(ae["rl_names"])                              # slot -> id
        self.rl_id_list = sorted(self.rl_names.values())                  # initial_vehicles.get_rl_ids()
        self.slot_of = {v: k for k, v in self.rl_names.items()}
        self.readd = [[] for _ in range(self.R)]
        self.ae = ae

    # ---- network kernel (network/traci.py:347-359 over A7's connection data)
    def next_edge(self, edge, lane):
        k = self.path.index(edge)
        if k + 1 >= len(self.path):
            return []
        nxt = self.path[k + 1]
        if edge[0] == ':' and nxt in self.conn:
            return [(nxt, self.conn[nxt][lane])]
        return [(nxt, lane)]

    def prev_edge(self, edge, lane):
        k = self.path.index(edge)
        if k == 0:
            return []
        prv = self.path[k - 1]
        if edge[0] != ':' and edge in self.conn:
            return [(prv, a) for a in sorted(self.conn[edge]) if self.conn[edge][a] == lane]
        return [(prv, lane)]

    # ---- vehicle kernel accessors on replica r (float64 views of the state, as the host reads it)
    def _edge_pos(self, r, i):
        x = float(self.x[r, i])
        edge = self.path[0]
        for e in self.path:
            if x >= self.e_start[e]:
                edge = e
        return edge, x - self.e_start[edge]

    def _lane(self, r, i):
        x = float(self.x[r, i])
        return int(self.route[r, i]) >> (int(x >= float(self.m1)) + int(x >= float(self.m2)))

    def _table_x(self, r, i):
        """get_x_by_id: Flow's edge-start table + position; internal edges resolve to their entry without it (O5)."""
        x = float(self.x[r, i])
        fx = 0.0
        for (st, it, fs, sl) in self.routes[0]["segments"]:
            if x >= float(st):
                fx = float(fs) + float(sl) * (x - float(st))
        return fx

    def _ids(self, r):
        """get_ids(): departure order."""
        alive = np.flatnonzero(self.route[r] >= 0)
        return [int(i) for i in alive[np.argsort(self.seq[r, alive], kind="stable")]]

    def _rl_ids(self, r):
        return sorted(self.rl_names[i] for i in self._ids(r) if i in self.rl_names)

    # ---- A1 + A2 + the simulator step + A5 + A6
    def step(self, actions=None):
        R = self.R
        n_cols = int(self.spec["num_rl"])
        acc = np.full((R, max(n_cols, 1)), np.nan, dtype=np.float64)
        targets = [dict() for _ in range(R)]
        tc = self.time_counter + int(self.spec.get("sims_per_step", 1))   # envs/base.py:324: counted before the actions
        if actions is not None:
            a = np.asarray(actions, dtype=np.float64).reshape(R, -1)
            if self.spec.get("clip_actions", True):                       # envs/base.py:599-615 clips with the Box
                lo = np.tile([-abs(float(self.ae["max_decel"])), -1.0], a.shape[1] // 2)
                hi = np.tile([float(self.ae["max_accel"]), 1.0], a.shape[1] // 2)
                a = np.clip(a, lo, hi)
            for r in range(R):
                rl_ids = self._rl_ids(r)
                num_rl = len(rl_ids)
                acceleration = a[r, ::2][:num_rl]
                direction = np.round(a[r, 1::2])[:num_rl]
                sorted_rl = sorted(rl_ids, key=lambda v: self._table_x(r, self.slot_of[v]))
                for k, vid in enumerate(sorted_rl):
                    i = self.slot_of[vid]
                    blocked = tc[r] <= float(self.ae["lane_change_duration"]) + float(self.h[r, i])
                    d = 0 if blocked else int(direction[k])
                    acc[r, self.veh[i]["rl_index"]] = acceleration[k]
                    lane = self._lane(r, i)
                    target = min(max(lane + d, 0), self.e_lanes[self._edge_pos(r, i)[0]] - 1)
                    if target != lane:
                        targets[r][i] = target
        if self.ae.get("add_rl_if_exit", True):
            for r in range(R):
                here = set(self._rl_ids(r))
                if len(here) != len(self.rl_id_list):
                    for vid in self.rl_id_list:
                        if vid not in here and vid not in self.readd[r]:
                            self.readd[r].append(vid)
        _, _, done = super().step(acc.astype(self.dt_) if actions is not None and n_cols > 0 else None)
        T = self.dt_.type
        for r in range(R):
            for i, target in targets[r].items():                         # A5
                if self.route[r, i] < 0:
                    continue
                x_i = float(self.x[r, i])
                g = int(x_i >= float(self.m1)) + int(x_i >= float(self.m2))
                if (int(self.route[r, i]) >> g) == target or target >= (self.P >> g):
                    continue
                clash = False
                if int(self.ae["lane_change_mode"].get(i, 0)) != 0:
                    for k in self._ids(r):
                        if k == i or (int(self.route[r, k]) >> g) != target:
                            continue
                        d = float(self.x[r, k]) - x_i
                        if (0 <= d < float(self.veh_len[k])) or (d < 0 and -d < float(self.veh_len[i])):
                            clash = True
                if not clash:
                    self.route[r, i] = target << g
            still = []
            for vid in self.readd[r]:                                     # A6
                i = self.slot_of[vid]
                if self.route[r, i] >= 0:
                    continue
                lane = self.rl_id_list.index(vid) % MAX_LANES * int(self.ae["scaling"])
                vs = self.veh[i]
                v_dep = min(float(vs.get("sumo_max_speed", 30.0)), float(self.spec["speed_limit"]))
                best = None
                for k in self._ids(r):
                    xk = float(self.x[r, k])
                    sh = int(max(xk, float(self.zip_d)) >= float(self.m1)) + int(max(xk, float(self.zip_d)) >= float(self.m2))
                    if (int(self.route[r, k]) >> sh) == (lane >> sh) and (best is None or xk < float(self.x[r, best])):
                        best = k
                ok = True
                if best is not None:
                    gap = (float(self.x[r, best]) - float(self.veh_len[best])) - 0.0
                    need = float(vs.get("sumo_min_gap", 2.5)) + max(
                        0.0, v_dep * float(vs.get("sumo_tau", 1.0)) + v_dep * (v_dep - float(self.v[r, best])) /
                        (2.0 * np.sqrt(float(vs["max_accel"]) * float(vs["max_decel"]))))
                    ok = gap >= need
                if not ok:
                    still.append(vid)
                    continue
                self.x[r, i], self.v[r, i], self.prev_v[r, i] = T(0.0), T(v_dep), T(0)
                self.last_accel[r, i] = T(0)
                self.route[r, i] = lane
                self.vmax[r, i] = T(vs.get("sumo_max_speed", 30.0))
                self.last_lc[r, i] = -(1 << 30)
                self.seq[r, i] = self.seq_ctr[r]
                self.seq_ctr[r] += 1
                self.origin[r, i] = -1 - i
            self.readd[r] = still
        self._update_neighbours(np.ones(R, dtype=bool))                   # TraCIVehicle.update after the step
        obs = [self.accel_state(r) for r in range(R)]
        rew = np.array([self.accel_reward(r, None if actions is None else a[r]) for r in range(R)])
        return obs, rew, done

    # ---- A3
    def _multi_lane(self, r, me, edge_dict):
        """_multi_lane_headways_util (vehicle/traci.py:804-867) for the vehicle in slot ``me``."""
        this_edge, this_pos = self._edge_pos(r, me)
        this_lane = self._lane(r, me)
        num_lanes = self.e_lanes[this_edge]
        headway, tailway = [1000] * num_lanes, [1000] * num_lanes
        leader, follower = [""] * num_lanes, [""] * num_lanes
        num_edges = len(self.path)
        for lane in range(num_lanes):
            if len(edge_dict[this_edge][lane]) > 0:
                ids, positions = zip(*edge_dict[this_edge][lane])
                ids, positions = list(ids), list(positions)
                index = bisect_left(positions, this_pos)
                if (lane == this_lane and index < len(positions) - 1) or (lane != this_lane and index < len(positions)):
                    if ids[index] == me:
                        leader[lane] = ids[index + 1]
                        headway[lane] = positions[index + 1] - this_pos - float(self.veh_len[leader[lane]])
                    else:
                        leader[lane] = ids[index]
                        headway[lane] = positions[index] - this_pos - float(self.veh_len[leader[lane]])
                if index > 0:
                    follower[lane] = ids[index - 1]
                    tailway[lane] = this_pos - positions[index - 1] - float(self.veh_len[me])
            if leader[lane] == "":                                        # _next_edge_leaders (:869-909)
                edge, ln, add_length = this_edge, lane, 0
                for _ in range(num_edges):
                    if len(self.next_edge(edge, ln)) == 0:
                        break
                    add_length += self.e_len[edge]
                    edge, ln = self.next_edge(edge, ln)[0]
                    if edge in edge_dict and len(edge_dict[edge][ln]) > 0:
                        leader[lane] = edge_dict[edge][ln][0][0]
                        headway[lane] = edge_dict[edge][ln][0][1] - this_pos + add_length - \
                            float(self.veh_len[leader[lane]])
                    if leader[lane] != "":
                        break
            if follower[lane] == "":                                      # _prev_edge_followers (:911-950)
                edge, ln, add_length = this_edge, lane, 0
                for _ in range(num_edges):
                    if len(self.prev_edge(edge, ln)) == 0:
                        break
                    edge, ln = self.prev_edge(edge, ln)[0]
                    add_length += self.e_len[edge]
                    if edge in edge_dict and len(edge_dict[edge][ln]) > 0:
                        tailway[lane] = this_pos - edge_dict[edge][ln][-1][1] + add_length - float(self.veh_len[me])
                        follower[lane] = edge_dict[edge][ln][-1][0]
                    if follower[lane] != "":
                        break
        return headway, tailway, leader, follower

    def accel_state(self, r):
        ids = self._ids(r)
        max_lanes = max(self.e_lanes.values())
        edge_dict = {}
        for i in ids:                                                     # vehicle/traci.py:727-745
            edge, pos = self._edge_pos(r, i)
            edge_dict.setdefault(edge, [[] for _ in range(max_lanes)])[self._lane(r, i)].append((i, pos))
        for edge in edge_dict:
            for lane in range(max_lanes):
                edge_dict[edge][lane].sort(key=lambda t: t[1])
        max_speed, scaling = float(self.ae["max_speed"]), int(self.ae["scaling"])
        num_rl = len(self.rl_id_list)
        rl_ids = self._rl_ids(r)
        rl_obs, id_counter = np.empty(0), 0
        for vid in rl_ids:
            num = self.rl_id_list.index(vid)
            if num != id_counter:
                rl_obs = np.concatenate((rl_obs, np.zeros(4 * (num - id_counter))))
                id_counter = num + 1
            else:
                id_counter += 1
            i = self.slot_of[vid]
            edge = self._edge_pos(r, i)[0]
            edge_num = -1 if edge[0] == ':' else int(edge) / 6
            rl_obs = np.concatenate((rl_obs, [self._table_x(r, i) / 1000, float(self.v[r, i]) / max_speed,
                                              self._lane(r, i) / MAX_LANES, edge_num]))
        diff = num_rl - int(rl_obs.shape[0] / 4)
        if diff > 0:
            rl_obs = np.concatenate((rl_obs, np.zeros(4 * diff)))
        rel, id_counter = np.empty(0), 0
        for vid in rl_ids:
            num = self.rl_id_list.index(vid)
            if num != id_counter:
                rel = np.concatenate((rel, np.zeros(4 * MAX_LANES * scaling * (num - id_counter))))
                id_counter = num + 1
            else:
                id_counter += 1
            n = MAX_LANES * scaling
            headway = np.asarray([1000] * n) / 1000
            tailway = np.asarray([1000] * n) / 1000
            vel_in_front = np.asarray([0] * n) / max_speed
            vel_behind = np.asarray([0] * n) / max_speed
            hw, tw, ld, fl = self._multi_lane(r, self.slot_of[vid], edge_dict)
            headway[0:len(hw)] = np.asarray(hw) / 1000
            tailway[0:len(tw)] = np.asarray(tw) / 1000
            for k, lead in enumerate(ld):
                if lead != '':
                    vel_in_front[k] = float(self.v[r, lead]) / max_speed
            for k, foll in enumerate(fl):                                 # (:608: `if lane_followers != ''`)
                vel_behind[k] = (float(self.v[r, foll]) if foll != '' else -1001) / max_speed
            rel = np.concatenate((rel, headway, tailway, vel_in_front, vel_behind))
        diff = num_rl - int(rel.shape[0] / (4 * MAX_LANES))
        if diff > 0:
            rel = np.concatenate((rel, np.zeros(4 * MAX_LANES * diff)))
        edge_obs = []
        for edge in self.ae["edge_list"]:
            veh = [i for i in ids if self._edge_pos(r, i)[0] == edge]
            if len(veh) > 0:
                avg = (sum(float(self.v[r, i]) for i in veh) / len(veh)) / max_speed
                edge_obs += [avg, len(veh) / self.e_len[edge]]
            else:
                edge_obs += [0, 0]
        return np.concatenate((rl_obs, rel, edge_obs))

    # ---- A4
    def accel_reward(self, r, actions):
        ids = self._ids(r)
        vel = np.array([float(self.v[r, i]) for i in ids])
        target = float(self.spec["target_velocity"])
        if any(vel < -100) or len(ids) == 0:                              # rewards.py:40-43 (fail: no crash ends this env)
            dv = 0.
        else:
            max_cost = np.linalg.norm(np.array([target] * len(ids)))
            cost = np.linalg.norm(vel - target)
            dv = max(max_cost - cost, 0) / (max_cost + np.finfo(np.float32).eps)
        rl = [float(self.v[r, self.slot_of[v]]) for v in self._rl_ids(r)]
        progress = np.linalg.norm(rl, 1) * 0.1 if rl else 0.0
        num_rl = len(rl)
        acts = np.zeros(0) if actions is None else np.asarray(actions, dtype=np.float64)
        penalty = 1.0 * np.sum(np.abs(np.round(acts[1::2])[:num_rl]))
        return dv + progress - penalty
