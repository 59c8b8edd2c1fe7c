"""Oracle restatement of the network geometry the hot path needs.

TEST INFRASTRUCTURE ONLY.  Follows flow/networks/ring.py,
flow/networks/figure_eight.py, flow/core/kernel/network/traci.py:178-212,
267-289 and flow/core/kernel/network/base.py:263-391, 515-608 literally.

The reference obtains edge lengths by running SUMO's ``netconvert`` (absent
here); the internal-edge (junction) lengths are therefore parameters.  The
default 0.1 m is the value Flow's own edge-start tables assume
(networks/ring.py:197, 209) and the value netconvert produced for one-lane
junctions in the reference fixture tests/fast_tests/test_files/fig8_test.net.xml.
"""
from math import pi

import numpy as np

VEHICLE_LENGTH = 5          # flow/core/kernel/network/base.py:10


class OracleNetwork:
    """Edge tables + the geometry queries of TraCIKernelNetwork."""

    def __init__(self, edges, edgestarts, internal_edgestarts, speed_limit, lanes=1):
        # edges: dict id -> length (internal ids start with ':')
        self.edges = dict(edges)
        self.edge_list = [e for e in edges if e[0] != ':']           # traci.py:153-155
        self.edgestarts = list(edgestarts)
        self.internal_edgestarts = list(internal_edgestarts)
        self.internal_edgestarts_dict = dict(internal_edgestarts)
        self.total_edgestarts = sorted(self.edgestarts + self.internal_edgestarts,
                                       key=lambda t: t[1])           # traci.py:205-206
        self.total_edgestarts_dict = dict(self.total_edgestarts)
        self._speed = speed_limit
        self._lanes = lanes

    # -- traci.py:291-325
    def edge_length(self, e):
        return self.edges.get(e, -1001)

    def length(self):
        return sum(self.edges.values())                             # traci.py:210-212

    def non_internal_length(self):
        return sum(self.edges[e] for e in self.edge_list)           # traci.py:178-180

    def max_speed(self):
        return self._speed

    def num_lanes(self, e):
        return self._lanes

    def get_edge_list(self):
        return self.edge_list

    def get_edge(self, x):                                          # traci.py:267-271
        for (edge, start_pos) in reversed(self.total_edgestarts):
            if x >= start_pos:
                return edge, x - start_pos

    def get_x(self, edge, position):                                # traci.py:273-289
        if len(edge) == 0:
            return -1001
        if edge[0] == ':':
            try:
                return self.internal_edgestarts_dict[edge] + position
            except KeyError:
                edge_name = edge.rsplit('_', 1)[0]
                return self.total_edgestarts_dict.get(edge_name, -1001)
        return self.total_edgestarts_dict[edge] + position

    # -- base.py:515-608
    def _get_start_pos_util(self, min_gap, bunching, lanes_distribution, num_vehicles):
        min_gap = max(0, min_gap)
        bunching = max(0, bunching)
        max_lane = max(self.num_lanes(e) for e in self.edge_list)
        if lanes_distribution > max_lane:
            lanes_distribution = max_lane
        elif lanes_distribution < 1:
            lanes_distribution = 1
        distribution_length = sum(
            self.edge_length(e) * min(self.num_lanes(e), lanes_distribution)
            for e in self.edge_list if self.edge_length(e) > min_gap + VEHICLE_LENGTH)
        available_edges = [e for e in self.edge_list
                           if self.edge_length(e) > min_gap + VEHICLE_LENGTH]
        available_length = distribution_length - lanes_distribution * bunching - \
            num_vehicles * (min_gap + VEHICLE_LENGTH)
        if available_length < 0:
            raise ValueError('There is not enough space to place all vehicles in the network.')
        return min_gap, lanes_distribution, available_length, available_edges

    # -- base.py:263-391 (edges_distribution='all', equal lane counts)
    def gen_even_start_pos(self, num_vehicles, x0=0, min_gap=0, bunching=0,
                           lanes_distribution=float('inf'), perturbation=0, rng=None):
        min_gap, lanes_distr, available_length, available_edges = \
            self._get_start_pos_util(min_gap, bunching, lanes_distribution, num_vehicles)
        if num_vehicles == 0:
            return [], []
        increment = available_length / num_vehicles
        x = x0
        car_count = 0
        startpositions, startlanes = [], []
        internal = dict(self.internal_edgestarts)
        while car_count < num_vehicles:
            pos = self.get_edge(x)
            while pos[0] in internal.keys():                        # base.py:339-354
                edges = [t[0] for t in self.total_edgestarts]
                indx_edge = next(i for i, e in enumerate(edges) if e == pos[0])
                if indx_edge == len(edges) - 1:
                    next_edge_pos = self.total_edgestarts[0]
                else:
                    next_edge_pos = self.total_edgestarts[indx_edge + 1]
                x = next_edge_pos[1]
                pos = (next_edge_pos[0], 0)
            while pos[0] not in available_edges:                    # base.py:357-359
                x = (x + self.edge_length(pos[0])) % self.non_internal_length()
                pos = self.get_edge(x)
            for lane in range(min([self.num_lanes(pos[0]), lanes_distr])):   # base.py:372-378
                car_count += 1
                startpositions.append(pos)
                startlanes.append(lane)
                if car_count == num_vehicles:
                    break
            x = (x + increment + VEHICLE_LENGTH + min_gap) % self.non_internal_length()  # :380
        if perturbation > 0:                                        # base.py:384-389
            rng = rng or np.random
            for i in range(num_vehicles):
                perturb = rng.normal(0, perturbation)
                edge, pos = startpositions[i]
                pos = max(0, min(self.edge_length(edge), pos + perturb))
                startpositions[i] = (edge, pos)
        return startpositions, startlanes


def ring_network(length=230, lanes=1, speed_limit=30, junction_length=0.1):
    """flow/networks/ring.py:95-216 + the internal edges netconvert adds."""
    edgelen = length / 4.                                           # ring.py:100
    edges = {"bottom": edgelen, "right": edgelen, "top": edgelen, "left": edgelen,
             ":right_0": junction_length, ":top_0": junction_length,
             ":left_0": junction_length, ":bottom_0": junction_length}
    jl = 0.1                                                        # ring.py:197
    edgestarts = [("bottom", 0),
                  ("right", 0.25 * length + jl),
                  ("top", 0.5 * length + 2 * jl),
                  ("left", 0.75 * length + 3 * jl)]                 # ring.py:199-202
    internal = [(":right_0", 0.25 * length),
                (":top_0", 0.5 * length + jl),
                (":left_0", 0.75 * length + 2 * jl),
                (":bottom_0", length + 3 * jl)]                     # ring.py:211-214
    return OracleNetwork(edges, edgestarts, internal, speed_limit, lanes)


def figure_eight_network(radius_ring=30, lanes=1, speed_limit=30,
                         center_length=None, junction_length=0.1):
    """flow/networks/figure_eight.py:70-74, 225-263.

    ``center_length`` is the netconvert length of ``:center_*`` (9.40 m in the
    reference fixture fig8_test.net.xml; Flow's own table assumes
    2.9 + 3.3*lanes).
    """
    ring_edgelen = radius_ring * pi / 2.
    junction_len = 2.9 + 3.3 * lanes                                # figure_eight.py:73
    inner_space_len = 0.28                                          # figure_eight.py:74
    if center_length is None:
        center_length = junction_len
    r = radius_ring
    edges = {"bottom": r, "top": r, "upper_ring": 3 * ring_edgelen, "right": r,
             "left": r, "lower_ring": 3 * ring_edgelen,
             ":bottom_0": junction_length, ":top_0": junction_length,
             ":right_0": junction_length, ":left_0": junction_length,
             ":center_0": center_length, ":center_1": center_length}
    edgestarts = [
        ("bottom", inner_space_len),
        ("top", r + junction_len + inner_space_len),
        ("upper_ring", 2 * r + junction_len + 2 * inner_space_len),
        ("right", 2 * r + 3 * ring_edgelen + junction_len + 3 * inner_space_len),
        ("left", 3 * r + 3 * ring_edgelen + 2 * junction_len + 3 * inner_space_len),
        ("lower_ring", 4 * r + 3 * ring_edgelen + 2 * junction_len + 4 * inner_space_len)]
    internal = [
        (":bottom", 0),
        (":center_{}".format(lanes), r + inner_space_len),
        (":top", 2 * r + junction_len + inner_space_len),
        (":right", 2 * r + 3 * ring_edgelen + junction_len + 2 * inner_space_len),
        (":center_0", 3 * r + 3 * ring_edgelen + junction_len + 3 * inner_space_len),
        (":left", 4 * r + 3 * ring_edgelen + 2 * junction_len + 3 * inner_space_len),
        # the two aimsun entries of figure_eight.py:256-260 also sit in the table
        ('bottom_to_top', r + inner_space_len),
        ('right_to_left', junction_len + 3 * inner_space_len)]
    return OracleNetwork(edges, edgestarts, internal, speed_limit, lanes)
