"""k.network: static geometry queries and initial placement.

Stands where flow/core/kernel/network/{base,traci}.py stood, without the
netconvert subprocess: edge lengths come from the Network description, internal
(junction) edge lengths from ``junction_length``.  Host Python, as in the
reference -- placement runs once per construction, not on the step path.
"""
import logging
import random

import numpy as np

from flow_amd.utils.exceptions import FatalFlowError

VEHICLE_LENGTH = 5          # flow/core/kernel/network/base.py:10


class NetworkKernel(object):
    """Geometry of one network instance (flow/core/kernel/network/traci.py:90-228, 267-359)."""

    def __init__(self, network, junction_length=0.1, center_length=None):
        self.network = network
        self.orig_name = network.orig_name
        self.name = network.name
        self.junction_length = junction_length
        types = {t["id"]: t for t in (network.types or [])}
        self._edges = {}
        for e in network.edges:
            t = types.get(e.get("type"), {})
            self._edges[e["id"]] = {"length": float(e["length"]),
                                    "lanes": int(e.get("numLanes", t.get("numLanes", 1))),
                                    "speed": float(e.get("speed", t.get("speed", 30)))}
        first = next(iter(self._edges.values()))
        for eid, length in network.specify_internal_edges(junction_length, center_length):
            self._edges[eid] = {"length": float(length), "lanes": first["lanes"], "speed": first["speed"]}
        # lane-drop networks: an internal edge has one lane per connection through it = the lanes of the edge before it
        # (netconvert's internal lanes ":4_0_0 .. :4_0_3" of the connections 3_i -> 4_floor(i/2), bottleneck.py:179-201)
        paths = network.specify_open_routes()
        self._drop_path = paths[0] if paths is not None and network.specify_lane_joins() is not None else None
        if self._drop_path is not None:
            for k, eid in enumerate(self._drop_path):
                if eid[0] == ':' and k > 0:
                    self._edges[eid]["lanes"] = self._edges[self._drop_path[k - 1]]["lanes"]
        self._edge_list = [e for e in self._edges if e[0] != ':']
        self._junction_list = [e for e in self._edges if e[0] == ':']
        self.edgestarts = list(network.edge_starts) if network.edge_starts is not None else None
        if self.edgestarts is None:                                  # traci.py:188-196
            length, self.edgestarts = 0, []
            for eid in sorted(self._edge_list):
                self.edgestarts.append((eid, length))
                length += self._edges[eid]['length']
        self.internal_edgestarts = list(network.internal_edge_starts)
        self.internal_edgestarts_dict = dict(self.internal_edgestarts)
        self.total_edgestarts = sorted(self.edgestarts + self.internal_edgestarts, key=lambda t: t[1])
        self.total_edgestarts_dict = dict(self.total_edgestarts)
        self.rts = network.routes
        # loop coordinate (what the simulator integrates) when it is not the table coordinate
        order = network.specify_loop_order()
        self.loop_starts = None
        if order is not None:
            self.loop_starts, s0 = [], 0.0
            for eid in order:
                self.loop_starts.append((eid, s0))
                s0 += self._edges[eid]['length']

    # ---- open networks (MergeNetwork): two routes sharing one coordinate, merge point at merge_x
    def open_tables(self):
        """Route tables of an open network for the simulator (None for closed networks): both routes are
        laid on one coordinate so that their common last edges coincide; ``segments`` rows are
        (start, internal, flow_start, flow_slope) with Flow's table coordinate of a point = flow_start +
        flow_slope * (x - start) (get_x; internal edges without a table entry have slope 0)."""
        paths = self.network.specify_open_routes()
        if paths is None:
            return None
        if len(paths) == 1:
            return self._lane_drop_tables(paths[0])
        a, b = paths
        n_common = 0
        while n_common < min(len(a), len(b)) and a[-1 - n_common] == b[-1 - n_common]:
            n_common += 1
        up = [sum(self._edges[e]['length'] for e in p[:len(p) - n_common]) for p in paths]
        merge_x = max(up)
        routes, self._open_starts = [], []
        for p, u in zip(paths, up):
            x, segs, starts = merge_x - u, [], []
            for e in p:
                if e[0] == ':' and e not in self.internal_edgestarts_dict:
                    segs.append((x, True, float(self.get_x(e, 0.0)), 0.0))            # traci.py:283-287
                else:
                    segs.append((x, e[0] == ':', float(self.get_x(e, 0.0)), 1.0))
                starts.append((e, x))
                x += self._edges[e]['length']
            routes.append(dict(start=merge_x - u, segments=segs))
            self._open_starts.append(starts)
        last_internal = paths[0][len(paths[0]) - n_common - 1]
        box_in = merge_x - self._edges[last_internal]['length'] if last_internal[0] == ':' else merge_x
        end_x = merge_x + sum(self._edges[e]['length'] for e in a[len(a) - n_common:])
        return dict(routes=routes, merge_x=merge_x, box_in=box_in, end_x=end_x, net_length=self.length())

    def _lane_drop_tables(self, path):
        """Lane-drop network (BottleneckNetwork): every lane drives the same edges; lanes 2q / 2q+1 join at the
        start of the edges ``specify_lane_joins`` names."""
        x, segs, starts = 0.0, [], []
        for e in path:
            if e[0] == ':':
                segs.append((x, True, float(self.get_x(e, 0.0)), 0.0))
            else:
                segs.append((x, False, float(self.get_x(e, 0.0)), 1.0))
            starts.append((e, x))
            x += self._edges[e]['length']
        self._open_starts = [starts]
        joins = self.network.specify_lane_joins()
        sd = dict(starts)
        last_internal = path[path.index(joins[1]) - 1]
        return dict(routes=[dict(start=0.0, segments=segs)], num_paths=self._edges[path[0]]['lanes'],
                    merge1_x=sd[joins[0]], merge2_x=sd[joins[1]], merge_x=sd[joins[1]],
                    box_in=sd[joins[1]] - self._edges[last_internal]['length'], end_x=x, net_length=self.length())

    def open_lane(self, route, x):
        """Lane index on its current edge of a vehicle that entered on lane ``route`` (lane-drop networks)."""
        t = getattr(self, "_lane_joins_x", None)
        if t is None:
            sd = dict(self._open_starts[0])
            t = self._lane_joins_x = [sd[e] for e in (self.network.specify_lane_joins() or [])]
        return int(route) >> sum(1 for m in t if x >= m)

    def open_locate(self, route, x):
        """(edge, position on it) of coordinate ``x`` on ``route`` of an open network."""
        starts = self._open_starts[route if len(self._open_starts) > 1 else 0]
        for (edge, start) in reversed(starts):
            if x >= start:
                return edge, x - start
        return starts[0][0], 0.0

    def open_coordinate(self, edge, position):
        """(route, x) of a point on ``edge`` (an edge both routes share belongs to route 0)."""
        for r, starts in enumerate(self._open_starts):
            for e, start in starts:
                if e == edge:
                    return r, start + position
        raise KeyError(edge)

    def locate(self, s):
        """(edge, position on it) of loop coordinate ``s``."""
        if self.loop_starts is None:
            return self.get_edge(s)
        for (edge, start) in reversed(self.loop_starts):
            if s >= start:
                return edge, s - start

    def loop_coordinate(self, edge, position):
        if self.loop_starts is None:
            return self.get_x(edge, position)
        return dict(self.loop_starts)[edge] + position

    def loop_segments(self):
        """[(start, internal, flow_start, flow_slope)] for fs_config.segments (None for a plain ring)."""
        if self.loop_starts is None:
            return None
        segs = []
        for edge, start in self.loop_starts:
            if edge[0] == ':' and edge not in self.internal_edgestarts_dict:
                segs.append((start, True, float(self.get_x(edge, 0.0)), 0.0))        # traci.py:283-287
            else:
                segs.append((start, edge[0] == ':', float(self.get_x(edge, 0.0)), 1.0))
        return segs

    def crossing_model(self, vehicle_length=5.0, half_width=0.9, time_gap=3.0):
        """fs_config.junction of a self-crossing loop (docs/HISTORY.md S-J), None otherwise."""
        cr = self.network.specify_crossing()
        if cr is None or self.loop_starts is None:
            return None
        starts = dict(self.loop_starts)
        a, b = cr
        a_in, b_in = starts[a], starts[b]
        ca, cb = self._edges[a]['length'], self._edges[b]['length']
        approach = [e for e, _ in self.loop_starts]
        look = self._edges[approach[approach.index(b) - 1]]['length']
        return dict(a_in=a_in, a_out=a_in + ca, b_in=b_in, b_out=b_in + cb, lookahead=look, time_gap=time_gap,
                    za_lo=a_in + ca / 2 - half_width, za_hi=a_in + ca / 2 + vehicle_length + half_width,
                    zb_lo=b_in + cb / 2 - half_width, zb_hi=b_in + cb / 2 + vehicle_length + half_width)

    # ---- static queries (traci.py:267-359)
    def get_edge(self, x):
        for (edge, start_pos) in reversed(self.total_edgestarts):
            if x >= start_pos:
                return edge, x - start_pos

    def get_x(self, edge, position):
        if len(edge) == 0:
            return -1001
        if edge[0] == ':':
            try:
                return self.internal_edgestarts_dict[edge] + position
            except KeyError:
                return self.total_edgestarts_dict.get(edge.rsplit('_', 1)[0], -1001)
        return self.total_edgestarts_dict[edge] + position

    def edge_length(self, edge_id):
        try:
            return self._edges[edge_id]['length']
        except KeyError:
            print('Error in edge length with key', edge_id)
            return -1001

    def length(self):
        return sum(e['length'] for e in self._edges.values())

    def non_internal_length(self):
        return sum(self._edges[e]['length'] for e in self._edge_list)

    def speed_limit(self, edge_id):
        try:
            return self._edges[edge_id]['speed']
        except KeyError:
            print('Error in speed limit with key', edge_id)
            return -1001

    def num_lanes(self, edge_id):
        try:
            return self._edges[edge_id]['lanes']
        except KeyError:
            print('Error in num lanes with key', edge_id)
            return -1001

    def max_speed(self):
        return max(self.speed_limit(e) for e in self._edge_list)

    def get_edge_list(self):
        return self._edge_list

    def get_junction_list(self):
        return self._junction_list

    def _drop_lane_map(self, edge_to):
        """{fromLane: toLane} of the connections into ``edge_to`` (specify_connections), None: lane i -> lane i."""
        conn = (self.network.connections or {}).get(edge_to) if isinstance(self.network.connections, dict) else None
        if not conn:
            return None
        return {int(c["fromLane"]): int(c["toLane"]) for c in conn}

    def next_edge(self, edge, lane):
        """network/traci.py:347-352 over the connection data netconvert would write (:946-975): an edge leads to the
        internal lane of its connection ("via"), an internal lane to the connection's target lane."""
        if self._drop_path is not None:
            p = self._drop_path
            if edge not in p or p.index(edge) + 1 >= len(p) or not 0 <= lane < self._edges[edge]["lanes"]:
                return []
            nxt = p[p.index(edge) + 1]
            if edge[0] == ':':
                m = self._drop_lane_map(nxt)
                return [(nxt, lane if m is None else m[lane])]
            return [(nxt, lane)]
        order = [t[0] for t in self.total_edgestarts if t[0] in self._edges]
        if edge not in order:
            return []
        return [(order[(order.index(edge) + 1) % len(order)], lane)]

    def prev_edge(self, edge, lane):
        if self._drop_path is not None:
            p = self._drop_path
            if edge not in p or p.index(edge) == 0 or not 0 <= lane < self._edges[edge]["lanes"]:
                return []
            prv = p[p.index(edge) - 1]
            if edge[0] != ':':
                m = self._drop_lane_map(edge)
                if m is not None:                                  # the internal lanes that end on this lane
                    return [(prv, i) for i in sorted(m) if m[i] == lane]
            return [(prv, lane)]
        order = [t[0] for t in self.total_edgestarts if t[0] in self._edges]
        if edge not in order:
            return []
        return [(order[(order.index(edge) - 1) % len(order)], lane)]

    def update(self, reset):
        pass

    def close(self):
        pass

    # ---- initial placement (base.py:221-608)
    def generate_starting_positions(self, initial_config, num_vehicles=None):
        num_vehicles = num_vehicles or self.network.vehicles.num_vehicles
        if initial_config.spacing == 'uniform':
            return self.gen_even_start_pos(initial_config, num_vehicles)
        if initial_config.spacing == 'random':
            return self.gen_random_start_pos(initial_config, num_vehicles)
        if initial_config.spacing == 'custom':
            return self.network.gen_custom_start_pos(cls=self, net_params=self.network.net_params,
                                                     initial_config=initial_config, num_vehicles=num_vehicles)
        raise FatalFlowError('"spacing" argument in initial_config does not contain a valid option')

    def _start_pos_util(self, initial_config, num_vehicles):
        min_gap = max(0, initial_config.min_gap)
        bunching = initial_config.bunching
        if bunching < 0:
            logging.warning('"bunching" cannot be negative; setting to 0')
            bunching = 0
        if initial_config.edges_distribution == 'all':
            edges = self.get_edge_list()
        else:
            edges = list(initial_config.edges_distribution)
        max_lane = max(self.num_lanes(e) for e in edges)
        lanes_distribution = initial_config.lanes_distribution
        if lanes_distribution > max_lane:
            lanes_distribution = max_lane
        elif lanes_distribution < 1:
            logging.warning('"lanes_distribution" is too small; setting to 1')
            lanes_distribution = 1
        usable = [e for e in edges if self.edge_length(e) > min_gap + VEHICLE_LENGTH]
        distribution_length = sum(self.edge_length(e) * min(self.num_lanes(e), lanes_distribution)
                                  for e in usable)
        available_length = distribution_length - lanes_distribution * bunching - \
            num_vehicles * (min_gap + VEHICLE_LENGTH)
        if available_length < 0:
            raise FatalFlowError('There is not enough space to place all vehicles in the network.')
        return min_gap, lanes_distribution, available_length, usable

    def _per_edge_start_pos(self, generator, initial_config, num_vehicles):
        """edges_distribution = {edge: number of vehicles} (network/base.py:285-308, 410-433): every edge is filled on
        its own, in the order of the dict, the vehicles taking the positions in id order.  (The reference leaves
        ``initial_config.edges_distribution`` set to the last edge afterwards; the config is not modified here.)"""
        import copy
        dist = initial_config.edges_distribution
        total = sum(dist[k] for k in dist)
        assert num_vehicles == total, \
            'Number of vehicles in edges_distribution and the Vehicles class do not match: {}, {}'.format(num_vehicles, total)
        startpositions, startlanes = [], []
        for key in dist:
            cfg = copy.copy(initial_config)
            cfg.edges_distribution = [key]
            pos, lane = generator(cfg, dist[key])
            startpositions.extend(pos)
            startlanes.extend(lane)
        return startpositions, startlanes

    def gen_even_start_pos(self, initial_config, num_vehicles):
        if isinstance(initial_config.edges_distribution, dict):
            return self._per_edge_start_pos(self.gen_even_start_pos, initial_config, num_vehicles)
        min_gap, lanes_distr, available_length, available_edges = \
            self._start_pos_util(initial_config, num_vehicles)
        if num_vehicles == 0:
            return [], []
        increment = available_length / num_vehicles
        lanes = [self.num_lanes(e) for e in self.get_edge_list()]
        uneven_lanes = any(lanes[0] != n for n in lanes[1:])
        x, car_count = initial_config.x0, 0
        startpositions, startlanes = [], []
        ordered = [t[0] for t in self.total_edgestarts]
        while car_count < num_vehicles:
            pos = self.get_edge(x)
            while pos[0] in self.internal_edgestarts_dict:           # never start inside a junction
                nxt = self.total_edgestarts[(ordered.index(pos[0]) + 1) % len(ordered)]
                x, pos = nxt[1], (nxt[0], 0)
            while pos[0] not in available_edges:
                x = (x + self.edge_length(pos[0])) % self.non_internal_length()
                pos = self.get_edge(x)
            if uneven_lanes and pos[1] < VEHICLE_LENGTH:
                pos = (pos[0], VEHICLE_LENGTH)
                x += VEHICLE_LENGTH
                increment -= (VEHICLE_LENGTH * self.num_lanes(pos[0])) / (num_vehicles - car_count)
            for lane in range(min([self.num_lanes(pos[0]), lanes_distr])):
                car_count += 1
                startpositions.append(pos)
                startlanes.append(lane)
                if car_count == num_vehicles:
                    break
            x = (x + increment + VEHICLE_LENGTH + min_gap) % self.non_internal_length()
        if initial_config.perturbation > 0:
            for i in range(num_vehicles):
                perturb = np.random.normal(0, initial_config.perturbation)
                edge, pos = startpositions[i]
                startpositions[i] = (edge, max(0, min(self.edge_length(edge), pos + perturb)))
        return startpositions, startlanes

    def gen_random_start_pos(self, initial_config, num_vehicles):
        if isinstance(initial_config.edges_distribution, dict):
            return self._per_edge_start_pos(self.gen_random_start_pos, initial_config, num_vehicles)
        min_gap, lanes_distr, available_length, available_edges = \
            self._start_pos_util(initial_config, num_vehicles)
        efs = min_gap + VEHICLE_LENGTH
        for edge in available_edges:
            available_length -= efs * min([self.num_lanes(edge), lanes_distr])
        init_absolute_pos = sorted(random.random() * available_length for _ in range(num_vehicles))
        for i in range(num_vehicles):
            init_absolute_pos[i] += (VEHICLE_LENGTH + min_gap) * i
        decrement, edge_indx = 0, 0
        startpositions, startlanes = [], []
        for i in range(num_vehicles):
            def locate():
                edge_i = available_edges[edge_indx]
                span = self.edge_length(edge_i) - efs
                pos_i = (init_absolute_pos[i] - decrement) % span
                lane_i = int(((init_absolute_pos[i] - decrement) - pos_i) / span)
                return edge_i, pos_i + efs, lane_i
            edge_i, pos_i, lane_i = locate()
            while lane_i > min([self.num_lanes(edge_i), lanes_distr]) - 1:
                decrement += min([self.num_lanes(edge_i), lanes_distr]) * (self.edge_length(edge_i) - efs)
                edge_indx += 1
                edge_i, pos_i, lane_i = locate()
            startpositions.append((edge_i, pos_i))
            startlanes.append(lane_i)
        return startpositions, startlanes
