"""k.vehicle: per-replica read view over the simulator state in HBM, plus the command sink.

Stands where flow/core/kernel/vehicle/{base,traci}.py stood.  The reference
rebuilds a dict-of-dicts from TraCI subscriptions every step
(vehicle/traci.py:119-259); here the state lives on the GPU as [R, N] arrays and
this object copies what is asked for, lazily, once per step.
"""
import numpy as np

from flow_amd import _lib as L
from flow_amd.controllers import RLController, SimCarFollowingController, SimLaneChangeController


class VehicleKernel(object):
    """View of replica ``replica`` of a FlowSim."""

    def __init__(self, master_kernel, sim_params):
        self.master_kernel = master_kernel
        self.sim_step = sim_params.sim_step
        self.sim = None
        self.replica = 0
        self.__ids, self.__human_ids, self.__controlled_ids = [], [], []
        self.__controlled_lc_ids, self.__rl_ids = [], []
        self.__vehicles = {}
        self.type_parameters, self.minGap = {}, {}
        self.num_vehicles = 0
        self.num_rl_vehicles = 0
        self._slot = {}
        self._cache = {}
        self._pending = None          # RL accelerations buffered by apply_acceleration
        self._pending_lc = None       # RL lane-change directions buffered by apply_lane_change
        self._observed = set()
        # open networks: the vehicles in the network change every step (vehicle/traci.py:145-216)
        self._open = False
        self._slot_id = {}
        self._num_departed, self._num_arrived = [], []
        self._departed_ids, self._arrived_ids, self._arrived_rl_ids = [], [], []

    # ---- construction (vehicle/traci.py:91-117, 261-372)
    def initialize(self, vehicles):
        self.type_parameters = vehicles.type_parameters
        self.minGap = vehicles.minGap
        self._open, self._slot_id, self._slot, self.sim = False, {}, {}, None    # until attach()
        self.__ids, self.__human_ids, self.__controlled_ids = [], [], []
        self.__controlled_lc_ids, self.__rl_ids = [], []
        self.__vehicles = {}
        for typ in vehicles.initial:
            for i in range(typ['num_vehicles']):
                veh_id = '{}_{}'.format(typ['veh_id'], i)
                tp = self.type_parameters[typ['veh_id']]
                acc_cls, acc_kw = tp["acceleration_controller"]
                lc_cls, lc_kw = tp["lane_change_controller"]
                rt = tp["routing_controller"]
                self.__vehicles[veh_id] = {
                    "type": typ['veh_id'], "initial_speed": typ['initial_speed'],
                    "acc_controller": acc_cls(veh_id, car_following_params=tp["car_following_params"],
                                              **(acc_kw or {})),
                    "lane_changer": lc_cls(veh_id=veh_id, **(lc_kw or {})),
                    "router": rt[0](veh_id=veh_id, router_params=rt[1]) if rt is not None else None,
                    "length": tp.get("length", 5.0)}
                self._slot[veh_id] = len(self.__ids)
                self.__ids.append(veh_id)
                if acc_cls == RLController:
                    self.__rl_ids.append(veh_id)
                else:
                    self.__human_ids.append(veh_id)
                    if acc_cls != SimCarFollowingController:
                        self.__controlled_ids.append(veh_id)
                    if lc_cls != SimLaneChangeController:
                        self.__controlled_lc_ids.append(veh_id)
        self.__rl_ids.sort()                                         # vehicle/traci.py:259, 366
        self.num_vehicles = len(self.__ids)
        self.num_rl_vehicles = len(self.__rl_ids)
        self._order = list(self.__ids)                               # id of the vehicle in slot i (ring order)

    def set_slot_order(self, order):
        """InitialConfig.shuffle (envs/base.py:268-292): the start positions were handed out in shuffled id order,
        so slot i (i-th position along the loop) holds ``order[i]``; get_ids() keeps its order."""
        assert sorted(order) == sorted(self.__ids)
        self._order = list(order)
        self._slot = {v: i for i, v in enumerate(order)}

    def attach(self, sim, replica=0):
        self.sim, self.replica = sim, replica
        self._cache = {}
        self._open = bool(getattr(sim, "open_net", False))
        if self._open:
            self._init_id_of_slot = {i: v for v, i in sim.spec["init_slot"].items()}
            self._refresh_open(reset=True)

    def update(self, reset):
        """Called after every simulation step: drop the host copies (vehicle/traci.py:119)."""
        self._cache = {}
        self._pending = None
        self._pending_lc = None
        if self._open:
            self._refresh_open(reset)

    def _sid(self, i):
        """id of the vehicle in slot i."""
        return self._slot_id[i] if self._open else self._order[i]

    def _new_vehicle(self, veh_id, type_name):
        tp = self.type_parameters[type_name]
        acc_cls, acc_kw = tp["acceleration_controller"]
        lc_cls, lc_kw = tp["lane_change_controller"]
        rt = tp["routing_controller"]
        return {"type": type_name, "initial_speed": tp.get("initial_speed", 0),
                "acc_controller": acc_cls(veh_id, car_following_params=tp["car_following_params"], **(acc_kw or {})),
                "lane_changer": lc_cls(veh_id=veh_id, **(lc_kw or {})),
                "router": rt[0](veh_id=veh_id, router_params=rt[1]) if rt is not None else None,
                "length": tp.get("length", 5.0)}

    def _refresh_open(self, reset):
        """Rebuild the id lists from the slot state on the device (TraCIVehicle.update, vehicle/traci.py:145-216):
        ids in departure order, arrived vehicles dropped, departed ones appended; SUMO names the k-th vehicle
        of InFlows entry "flow_f" as "flow_f.k"."""
        spec = self.sim.spec
        route = self._field(L.FS_FIELD_ROUTE)
        seq = self._field(L.FS_FIELD_SEQ)
        origin = self._field(L.FS_FIELD_ORIGIN)
        names, base, caps = spec["slot_types"], spec["slot_base"], spec["slot_caps"]
        slots = sorted((int(seq[i]), i) for i in range(len(route)) if route[i] >= 0)
        old_ids, old_rl = list(self.__ids), list(self.__rl_ids)
        ids, slot_of, id_of = [], {}, {}
        for _, i in slots:
            o = int(origin[i])
            veh_id = self._init_id_of_slot[i] if o < 0 else "%s.%d" % (spec["inflows"][o >> 20]["name"], o & 0xFFFFF)
            ids.append(veh_id)
            slot_of[veh_id], id_of[i] = i, veh_id
        vehicles = {}
        for veh_id in ids:
            if veh_id in self.__vehicles:
                vehicles[veh_id] = self.__vehicles[veh_id]
            else:
                i = slot_of[veh_id]
                tname = next(n for n in names if base[n] <= i < base[n] + caps[n])
                vehicles[veh_id] = self._new_vehicle(veh_id, tname)
        self.__vehicles = vehicles
        self.__ids, self._slot, self._slot_id = ids, slot_of, id_of
        self.__rl_ids, self.__human_ids, self.__controlled_ids, self.__controlled_lc_ids = [], [], [], []
        for veh_id in ids:
            tp = self.type_parameters[vehicles[veh_id]["type"]]
            if tp["acceleration_controller"][0] == RLController:
                self.__rl_ids.append(veh_id)
            else:
                self.__human_ids.append(veh_id)
                if tp["acceleration_controller"][0] != SimCarFollowingController:
                    self.__controlled_ids.append(veh_id)
                if tp["lane_change_controller"][0] != SimLaneChangeController:
                    self.__controlled_lc_ids.append(veh_id)
        self.__rl_ids.sort()                                         # vehicle/traci.py:259
        self.num_vehicles, self.num_rl_vehicles = len(ids), len(self.__rl_ids)
        if reset:                                                    # vehicle/traci.py:184-195
            for h in (self._num_departed, self._num_arrived, self._departed_ids, self._arrived_ids,
                      self._arrived_rl_ids):
                del h[:]
            return
        now = set(ids)
        departed = [v for v in ids if v not in set(old_ids)]
        arrived = [v for v in old_ids if v not in now]
        self._num_departed.append(len(departed))
        self._num_arrived.append(len(arrived))
        self._departed_ids.append(departed)
        self._arrived_ids.append(arrived)
        self._arrived_rl_ids.append([v for v in arrived if v in old_rl])

    def reset(self):
        self._cache = {}

    # ---- id lists (vehicle/base.py:327-380)
    def get_ids(self):
        return self.__ids

    def get_human_ids(self):
        return self.__human_ids

    def get_controlled_ids(self):
        return self.__controlled_ids

    def get_controlled_lc_ids(self):
        return self.__controlled_lc_ids

    def get_rl_ids(self):
        return self.__rl_ids

    # vehicle/traci.py:493-533; one history entry per Env.step here (the reference appends one per sub-step)
    def get_arrived_ids(self):
        return self._arrived_ids[-1] if self._arrived_ids else 0

    def get_arrived_rl_ids(self):
        return self._arrived_rl_ids[-1] if self._arrived_rl_ids else 0

    def get_departed_ids(self):
        return self._departed_ids[-1] if self._departed_ids else 0

    def get_num_arrived(self):
        return self._num_arrived[-1] if self._num_arrived else 0

    def _rate(self, hist, time_span):
        if len(hist) == 0:
            return 0
        n_sub = int(self.sim.spec.get("sims_per_step", 1)) if self.sim is not None else 1
        steps = max(int(time_span / (self.sim_step * n_sub)), 1)
        window = hist[-steps:]
        return 3600 * sum(window) / (len(window) * n_sub * self.sim_step)

    def get_inflow_rate(self, time_span):
        return self._rate(self._num_departed, time_span)

    def get_outflow_rate(self, time_span):
        return self._rate(self._num_arrived, time_span)

    def set_observed(self, veh_id):
        self._observed.add(veh_id)

    def remove_observed(self, veh_id):
        self._observed.discard(veh_id)

    def get_observed_ids(self):
        return list(self._observed)

    # ---- state reads
    def _field(self, field):
        if field not in self._cache:
            a = self.sim.get_state(field)
            self._cache[field] = a[self.replica]
        return self._cache[field]

    def _ring_neighbour(self, i, step):
        """Slot ``step`` places ahead of slot i (closed single-lane loops keep their order)."""
        return (i + step) % self.num_vehicles

    def _vec(self, veh_id, fn, error=-1001):
        if isinstance(veh_id, (list, np.ndarray)):
            return [self._vec(v, fn, error) for v in veh_id]
        if veh_id not in self._slot:
            return error
        return fn(self._slot[veh_id])

    # ---- array views (no reference counterpart): what the host-side reward helpers reduce over
    def slots_of(self, veh_ids=None):
        """Slot indices (into the simulator's [N] state rows of this replica) of ``veh_ids`` (default: get_ids())."""
        ids = self.get_ids() if veh_ids is None else veh_ids
        return np.fromiter((self._slot[v] for v in ids if v in self._slot), dtype=np.int64)

    def speeds(self, veh_ids=None):
        """float64 array of the speeds of ``veh_ids`` (default: every vehicle in the network), one device read per step."""
        return self._field(L.FS_FIELD_VEL)[self.slots_of(veh_ids)].astype(np.float64)

    def _need_aux(self, what):
        """The previous-speed / acceleration fields are only kept current by handles created with track_aux (the scalar
        Env's default; off under SumoParams(precision='mixed') and VecFlowEnv's default): say so instead of returning the
        values of the last reset."""
        if not getattr(self.sim, "spec", {}).get("track_aux", False) and not getattr(self, "_warned_aux", False):
            import warnings
            self._warned_aux = True
            warnings.warn("k.vehicle.%s: this handle steps without the previous-speed / acceleration fields "
                          "(track_aux = 0: precision='mixed', or VecFlowEnv(track_aux=False)); the values returned are "
                          "those of the last reset" % what, stacklevel=3)

    def previous_speeds(self, veh_ids=None):
        self._need_aux("previous_speeds")
        return self._field(L.FS_FIELD_PREV_VEL)[self.slots_of(veh_ids)].astype(np.float64)

    def get_speed(self, veh_id, error=-1001):
        return self._vec(veh_id, lambda i: float(self._field(L.FS_FIELD_VEL)[i]), error)

    def get_default_speed(self, veh_id, error=-1001):
        return self.get_speed(veh_id, error)

    def get_previous_speed(self, veh_id, error=-1001):
        self._need_aux("get_previous_speed")
        return self._vec(veh_id, lambda i: float(self._field(L.FS_FIELD_PREV_VEL)[i]), error)

    def get_accel(self, veh_id, error=None):
        self._need_aux("get_accel")
        return self._vec(veh_id, lambda i: float(self._field(L.FS_FIELD_ACCEL)[i]), error)

    def get_x_by_id(self, veh_id):
        """Flow's absolute position: edge start of the network's table + position on the edge
        (vehicle/traci.py:1011-1017)."""
        net = self.master_kernel.network
        if net.loop_starts is None and not self._open:
            return self._vec(veh_id, lambda i: float(self._field(L.FS_FIELD_POS)[i]), 0.)
        return self._vec(veh_id, lambda i: float(net.get_x(*self._edge_pos(i))), 0.)

    def _edge_pos(self, i):
        if self._open:
            return self.master_kernel.network.open_locate(int(self._field(L.FS_FIELD_ROUTE)[i]),
                                                          float(self._field(L.FS_FIELD_POS)[i]))
        return self.master_kernel.network.locate(float(self._field(L.FS_FIELD_POS)[i]))

    def get_route(self, veh_id, error=None):
        """Edges of the vehicle's route (vehicle/traci.py:571-578)."""
        err = list() if error is None else error
        if not self._open:
            return self._vec(veh_id, lambda i: self.master_kernel.network.rts.get(self._edge_pos(i)[0], err), err)
        paths = self.master_kernel.network.network.specify_open_routes()
        return self._vec(veh_id, lambda i: [e for e in paths[int(self._field(L.FS_FIELD_ROUTE)[i])] if e[0] != ':'],
                         err)

    def get_edge(self, veh_id, error=""):
        return self._vec(veh_id, lambda i: self._edge_pos(i)[0], error)

    def get_position(self, veh_id, error=-1001):
        return self._vec(veh_id, lambda i: self._edge_pos(i)[1], error)

    def _multilane(self):
        return self.sim is not None and int(self.sim.spec.get("num_lanes", 1)) > 1

    def get_lane(self, veh_id, error=-1001):
        if self._open and self.sim.spec.get("network") == "bottleneck":      # lane on the current edge
            net = self.master_kernel.network
            return self._vec(veh_id, lambda i: net.open_lane(int(self._field(L.FS_FIELD_ROUTE)[i]),
                                                             float(self._field(L.FS_FIELD_POS)[i])), error)
        if self._multilane():
            return self._vec(veh_id, lambda i: int(self._field(L.FS_FIELD_LANE)[i]), error)
        return self._vec(veh_id, lambda i: 0, error)

    def get_headway(self, veh_id, error=-1001):
        return self._vec(veh_id, lambda i: float(self._field(L.FS_FIELD_HEADWAY)[i]), error)

    def get_leader(self, veh_id, error=""):
        n = self.num_vehicles
        if self._multilane() or self._open:
            def lead(i):
                j = int(self._field(L.FS_FIELD_LEADER)[i])
                return self._sid(j) if j >= 0 else None
            return self._vec(veh_id, lead, error)
        return self._vec(veh_id, lambda i: self._order[self._ring_neighbour(i, 1)] if n > 1 else None, error)

    def get_follower(self, veh_id, error=""):
        n = self.num_vehicles
        if self._open:                                   # the sticky entry of vehicle/traci.py:243-250
            def foll(i):
                j = int(self._field(L.FS_FIELD_FOLLOWER)[i])
                return self._sid(j) if j >= 0 else None
            return self._vec(veh_id, foll, error)
        if self._multilane():
            def foll(i):
                lead = self._field(L.FS_FIELD_LEADER)
                cand = [j for j in range(n) if int(lead[j]) == i]
                if not cand:
                    return None
                hw = self._field(L.FS_FIELD_HEADWAY)
                return self._order[min(cand, key=lambda j: hw[j])]     # vehicle/traci.py:243-250
            return self._vec(veh_id, foll, error)
        return self._vec(veh_id, lambda i: self._order[self._ring_neighbour(i, -1)] if n > 1 else None, error)

    def get_max_speed(self, veh_id, error=-1001):
        """maxSpeed of the SUMO car-following model of the vehicle (vehicle/traci.py get_max_speed)."""
        if not self._open:
            return self._vec(veh_id, lambda i: float(self.sim.spec["vehicles"][i].get("sumo_max_speed", 30.0)), error)
        return self._vec(veh_id, lambda i: float(self._field(L.FS_FIELD_MAX_SPEED)[i]), error)

    def set_max_speed(self, veh_id, max_speed):
        """vehicle/traci.py set_max_speed (setMaxSpeed): open networks only -- closed loops keep it per type."""
        if not self._open:
            raise NotImplementedError("set_max_speed is built for open networks")
        m = self.sim.get_state(L.FS_FIELD_MAX_SPEED)
        m[self.replica, self._slot[veh_id]] = max_speed
        self.sim.set_state(L.FS_FIELD_MAX_SPEED, m)
        self._cache.pop(L.FS_FIELD_MAX_SPEED, None)

    def get_length(self, veh_id, error=-1001):
        return self._vec(veh_id, lambda i: self.__vehicles[self._sid(i)]["length"], error)

    def get_type(self, veh_id):
        return self.__vehicles[veh_id]["type"]

    def get_initial_speed(self, veh_id, error=-1001):
        return self._vec(veh_id, lambda i: self.__vehicles[self._sid(i)]["initial_speed"], error)

    def get_acc_controller(self, veh_id, error=None):
        return self._vec(veh_id, lambda i: self.__vehicles[self._sid(i)]["acc_controller"], error)

    def get_lane_changing_controller(self, veh_id, error=None):
        return self._vec(veh_id, lambda i: self.__vehicles[self._sid(i)]["lane_changer"], error)

    def get_routing_controller(self, veh_id, error=None):
        return self._vec(veh_id, lambda i: self.__vehicles[self._sid(i)]["router"], error)

    def get_ids_by_edge(self, edges):
        if isinstance(edges, (list, np.ndarray)):
            return sum([self.get_ids_by_edge(e) for e in edges], [])
        if "by_edge" not in self._cache:                 # one pass per simulation step
            table = {}
            for v in self.__ids:
                table.setdefault(self.get_edge(v), []).append(v)
            self._cache["by_edge"] = table
        return list(self._cache["by_edge"].get(edges, []))

    def get_last_lc(self, veh_id, error=-1001):
        """This fork returns the HEADWAY here (vehicle/traci.py:604-614); kept unless the env was
        built with LAST_LC_QUIRK = False, in which case it is the time of the last lane change."""
        if isinstance(veh_id, (list, np.ndarray)):
            return [self.get_headway(v, error) for v in veh_id]               # vehicle/traci.py:606-607
        if veh_id not in self.__rl_ids:
            return error
        if self.sim is not None and not self.sim.spec.get("last_lc_quirk", True):
            t = int(self._field(L.FS_FIELD_LAST_LC)[self._slot[veh_id]])
            return -float("inf") if t < 0 else t
        return self.get_headway(veh_id, error)

    def _lane_neighbours(self, i):
        """Per lane: (leader id, headway, follower id, tailway) as _multi_lane_headways computes them
        (vehicle/traci.py:776-867): headway = pos_lead - pos - len(lead), tailway = pos - pos_follow - len(self),
        '' / 1000 for an empty lane.  The candidates of a lane are ALL its vehicles, this one included: the
        reference's walk over the edges ahead / behind ends on the vehicle's own edge, so alone in its lane it is its
        own leader and follower, one lap away.  (Host accessor for user code and rendering; the observation of
        LaneChangeAccelPOEnv is written by the kernel.)"""
        if self._open and self.sim.spec.get("network") == "bottleneck":
            return self._lane_neighbours_drop(i)
        lanes = int(self.sim.spec.get("num_lanes", 1))
        x = self._field(L.FS_FIELD_POS).astype(np.float64)
        ln = self._field(L.FS_FIELD_LANE) if lanes > 1 else np.zeros(self.num_vehicles, dtype=np.int32)
        lap = self.master_kernel.network.length()
        slots = np.arange(self.num_vehicles)
        ahead = x - x[i]
        ahead[(ahead < 0) | (slots == i)] += lap          # bisect_left: a vehicle at the same position is a leader
        behind = x[i] - x
        behind[behind <= 0] += lap
        out = []
        for lane in range(lanes):
            cand = slots[np.asarray(ln) == lane]
            if cand.size == 0:
                out.append(("", 1000, "", 1000))
                continue
            lj = cand[np.argmin(ahead[cand])]
            fj = cand[::-1][np.argmin(behind[cand[::-1]])]
            out.append((self._order[lj], ahead[lj] - self.__vehicles[self._order[lj]]["length"],
                        self._order[fj], behind[fj] - self.__vehicles[self._order[i]]["length"]))
        return out

    def _drop_geometry(self):
        """Static tables of the lane-drop network for the per-lane neighbour walk: edge starts on the route coordinate,
        joins passed at each edge, and the lengths the reference's walk accumulates (``add_length`` of
        vehicle/traci.py:898-906 / :940-943: summed edge by edge in walking order, so the same floating-point sums)."""
        g = getattr(self, "_drop_geo", None)
        if g is None:
            net = self.master_kernel.network
            path = net._drop_path
            starts = np.array([s for _, s in net._open_starts[0]], dtype=np.float64)
            sd = dict(net._open_starts[0])
            joins = [sd[e] for e in net.network.specify_lane_joins()]
            g_edge = np.array([sum(1 for m in joins if s >= m) for s in starts], dtype=np.int64)
            n = len(path)
            fwd, bwd = np.zeros((n, n)), np.zeros((n, n))
            for a in range(n):
                acc = 0.0
                for b in range(a + 1, n):                      # leader on edge b: lengths of edges a .. b-1
                    acc += net.edge_length(path[b - 1])
                    fwd[a, b] = acc
                acc = 0.0
                for b in range(a - 1, -1, -1):                 # follower on edge b: lengths of edges a-1 .. b
                    acc += net.edge_length(path[b])
                    bwd[a, b] = acc
            g = self._drop_geo = (path, starts, g_edge, fwd, bwd)
        return g

    def _lane_neighbours_drop(self, i):
        """_multi_lane_headways_util on the lane-drop network (vehicle/traci.py:776-950), in route coordinates.  For lane
        q of the vehicle's edge: the leader is the first vehicle at or ahead of it (bisect_left: a vehicle at the same
        position counts as ahead; ties in id-list order) in lane q of this edge, else the first vehicle of the lane the
        connections lead to on the edges ahead; the follower is the last vehicle strictly behind in lane q of this edge,
        else the last one on the edges behind, walking prev_edge(...)[0] -- the lowest of the lanes that join."""
        path, starts, g_edge, fwd, bwd = self._drop_geometry()
        x = self._field(L.FS_FIELD_POS).astype(np.float64)
        route = np.asarray(self._field(L.FS_FIELD_ROUTE)).astype(np.int64)
        seq = np.asarray(self._field(L.FS_FIELD_SEQ)).astype(np.int64)
        alive = route >= 0
        e = np.clip(np.searchsorted(starts, x, side="right") - 1, 0, len(path) - 1)
        pos = x - starts[e]
        lane = np.where(alive, route, 0) >> g_edge[e]
        length = np.array([self.__vehicles[self._slot_id[k]]["length"] if alive[k] else 0.0 for k in range(len(x))])
        others = alive.copy()
        others[i] = False
        ei, gi = int(e[i]), int(g_edge[e[i]])
        n_lanes = self.master_kernel.network.num_lanes(path[ei])
        out = []
        for q in range(n_lanes):
            up = g_edge[e] - gi                                          # joins between my edge and theirs (>= 0 ahead)
            ahead = others & (((e == ei) & (lane == q) & (pos >= pos[i])) |
                              ((e > ei) & (lane == (q >> np.maximum(up, 0)))))
            behind = others & (((e == ei) & (lane == q) & (pos < pos[i])) |
                               ((e < ei) & (lane == (q << np.maximum(-up, 0)))))
            lead, head, foll, tail = "", 1000, "", 1000
            if ahead.any():
                c = np.flatnonzero(ahead)
                k = c[np.lexsort((seq[c], pos[c], e[c]))[0]]            # nearest edge, then position, then id-list order
                lead = self._slot_id[k]
                head = pos[k] - pos[i] - length[k] if e[k] == ei else pos[k] - pos[i] + fwd[ei, e[k]] - length[k]
            if behind.any():
                c = np.flatnonzero(behind)
                k = c[np.lexsort((-seq[c], -pos[c], -e[c]))[0]]
                foll = self._slot_id[k]
                tail = pos[i] - pos[k] - length[i] if e[k] == ei else pos[i] - pos[k] + bwd[ei, e[k]] - length[i]
            out.append((lead, head, foll, tail))
        return out

    def lane_neighbour_table(self, veh_id):
        """[(leader id, headway, follower id, tailway)] per lane for one vehicle ('' / 1000 where a lane is empty)."""
        return self._lane_neighbours(self._slot[veh_id])

    def get_lane_leaders(self, veh_id, error=None):
        return self._vec(veh_id, lambda i: [t[0] for t in self._lane_neighbours(i)], error)

    def get_lane_headways(self, veh_id, error=None):
        return self._vec(veh_id, lambda i: [t[1] for t in self._lane_neighbours(i)], error)

    def get_lane_followers(self, veh_id, error=None):
        return self._vec(veh_id, lambda i: [t[2] for t in self._lane_neighbours(i)], error)

    def get_lane_tailways(self, veh_id, error=None):
        return self._vec(veh_id, lambda i: [t[3] for t in self._lane_neighbours(i)], error)

    def get_lane_leaders_speed(self, veh_id, error=None):
        return self._vec(veh_id, lambda i: [self.get_speed(t[0]) if t[0] else 0.0
                                            for t in self._lane_neighbours(i)], error)

    def get_lane_followers_speed(self, veh_id, error=None):
        return self._vec(veh_id, lambda i: [self.get_speed(t[2]) if t[2] else 0.0
                                            for t in self._lane_neighbours(i)], error)

    # ---- commands
    def apply_acceleration(self, veh_ids, acc):
        """Buffer RL accelerations for the next simulation step (vehicle/traci.py:952-963).
        Accelerations of non-RL vehicles come from their in-kernel controllers."""
        if isinstance(veh_ids, str):
            veh_ids, acc = [veh_ids], [acc]
        if self._pending is None:
            self._pending = {}
        for vid, a in zip(veh_ids, acc):
            if a is None or vid not in self._slot:
                continue
            if vid not in self.__rl_ids:
                raise NotImplementedError(
                    "apply_acceleration on a non-RL vehicle: its controller runs in the HIP kernel")
            self._pending[vid] = float(np.asarray(a, dtype=np.float64).reshape(-1)[0])

    def apply_lane_change(self, veh_ids, direction):
        """vehicle/traci.py:965-997: direction validation kept; single-lane networks clip to lane 0."""
        if isinstance(veh_ids, str):
            veh_ids, direction = [veh_ids], [direction]
        if any(d not in [-1, 0, 1] for d in direction):
            raise ValueError("Direction values for lane changes may only be: -1, 0, or 1.")
        if self._pending_lc is None:
            self._pending_lc = {}
        for vid, d in zip(veh_ids, direction):
            if d == 0 or vid not in self._slot:
                continue
            if vid not in self.__rl_ids:
                raise NotImplementedError("lane changes of non-RL vehicles are decided in the HIP kernel")
            self._pending_lc[vid] = int(d)

    def choose_routes(self, veh_ids, route_choices):
        pass

    # ---- test back-doors (vehicle/traci.py:411-425)
    def _rs(self, veh_id):
        i = self._slot[veh_id]
        return self.replica, i

    def test_set_speed(self, veh_id, speed):
        v = self.sim.get_state(L.FS_FIELD_VEL)
        v[self._rs(veh_id)] = speed
        self.sim.set_state(L.FS_FIELD_VEL, v)
        self._cache = {}

    def test_set_position(self, veh_id, x):
        p = self.sim.get_state(L.FS_FIELD_POS)
        p[self._rs(veh_id)] = x
        self.sim.set_state(L.FS_FIELD_POS, p)
        self._cache = {}
