"""FigureEightNetwork (flow/networks/figure_eight.py): two rings joined by a crossing; every vehicle drives
the closed route bottom -> top -> upper_ring -> right -> left -> lower_ring."""
import numpy as np
from numpy import pi, sin, cos, linspace

from flow_amd.core.params import InitialConfig, TrafficLightParams
from flow_amd.networks.base import Network

ADDITIONAL_NET_PARAMS = {
    # radius of the circular components
    "radius_ring": 30,
    # number of lanes
    "lanes": 1,
    # speed limit for all edges
    "speed_limit": 30,
    # resolution of the curved portions
    "resolution": 40
}

# length netconvert gives the ':center_*' internal edges of a one-lane figure eight in the reference's own
# fixture (tests/fast_tests/test_files/fig8_test.net.xml; pinned 9.40 at test_scenario_base_class.py:690-693)
CENTER_LENGTH_ONE_LANE = 9.4


class FigureEightNetwork(Network):
    """flow/networks/figure_eight.py:22-263."""

    def __init__(self, name, vehicles, net_params, initial_config=InitialConfig(),
                 traffic_lights=TrafficLightParams(), detector_params=None):
        for p in ADDITIONAL_NET_PARAMS.keys():
            if p not in net_params.additional_params:
                raise KeyError('Network parameter "{}" not supplied'.format(p))
        ring_radius = net_params.additional_params["radius_ring"]
        self.ring_edgelen = ring_radius * np.pi / 2.
        self.intersection_len = 2 * ring_radius
        self.junction_len = 2.9 + 3.3 * net_params.additional_params["lanes"]
        self.inner_space_len = 0.28
        super().__init__(name, vehicles, net_params, initial_config, traffic_lights, detector_params)

    def specify_nodes(self, net_params):
        r = net_params.additional_params["radius_ring"]
        return [{"id": "center", "x": 0, "y": 0, "radius": (2.9 + 3.3 * net_params.additional_params["lanes"]) / 2,
                 "type": "priority"},
                {"id": "right", "x": r, "y": 0, "type": "priority"}, {"id": "top", "x": 0, "y": r, "type": "priority"},
                {"id": "left", "x": -r, "y": 0, "type": "priority"},
                {"id": "bottom", "x": 0, "y": -r, "type": "priority"}]

    def specify_edges(self, net_params):
        r = net_params.additional_params["radius_ring"]
        resolution = net_params.additional_params["resolution"]
        ring_edgelen = 3 * r * pi / 2.
        half = r                                             # intersection_edgelen / 2
        edges = [{"id": "bottom", "type": "edgeType", "priority": "78", "from": "bottom", "to": "center", "length": half},
                 {"id": "top", "type": "edgeType", "priority": 78, "from": "center", "to": "top", "length": half},
                 {"id": "right", "type": "edgeType", "priority": 46, "from": "right", "to": "center", "length": half},
                 {"id": "left", "type": "edgeType", "priority": 46, "from": "center", "to": "left", "length": half}]
        edges += [{"id": "upper_ring", "type": "edgeType", "from": "top", "to": "right", "length": ring_edgelen,
                   "shape": [(r * (1 - cos(t)), r * (1 + sin(t))) for t in linspace(0, 3 * pi / 2, resolution)]},
                  {"id": "lower_ring", "type": "edgeType", "from": "left", "to": "bottom", "length": ring_edgelen,
                   "shape": [(-r + r * cos(t), -r + r * sin(t)) for t in linspace(pi / 2, 2 * pi, resolution)]}]
        return edges

    def specify_types(self, net_params):
        return [{"id": "edgeType", "numLanes": net_params.additional_params["lanes"],
                 "speed": net_params.additional_params["speed_limit"]}]

    def specify_routes(self, net_params):
        loop = ["bottom", "top", "upper_ring", "right", "left", "lower_ring"]
        return {e: loop[i:] + loop[:i] for i, e in enumerate(loop)}

    def specify_connections(self, net_params):
        lanes = net_params.additional_params["lanes"]
        conn = []
        for i in range(lanes):
            conn += [{"from": "bottom", "to": "top", "fromLane": str(i), "toLane": str(i)}]
            conn += [{"from": "right", "to": "left", "fromLane": str(i), "toLane": str(i)}]
        return {"center": conn}

    def specify_edge_starts(self):
        il, jl, sp, re = self.intersection_len, self.junction_len, self.inner_space_len, self.ring_edgelen
        return [("bottom", sp), ("top", il / 2 + jl + sp), ("upper_ring", il + jl + 2 * sp),
                ("right", il + 3 * re + jl + 3 * sp), ("left", 3 / 2 * il + 3 * re + 2 * jl + 3 * sp),
                ("lower_ring", 2 * il + 3 * re + 2 * jl + 4 * sp)]

    def specify_internal_edge_starts(self):
        il, jl, sp, re = self.intersection_len, self.junction_len, self.inner_space_len, self.ring_edgelen
        lanes = self.net_params.additional_params['lanes']
        return [(":bottom", 0), (":center_{}".format(lanes), il / 2 + sp), (":top", il + jl + sp),
                (":right", il + 3 * re + jl + 2 * sp), (":center_0", 3 / 2 * il + 3 * re + jl + 3 * sp),
                (":left", 2 * il + 3 * re + 2 * jl + 3 * sp),
                ('bottom_to_top', il / 2 + sp), ('right_to_left', + jl + 3 * sp)]      # aimsun entries, :256-260

    # ---- what netconvert would add (no netconvert here)
    def specify_internal_edges(self, junction_length, center_length=None):
        c = CENTER_LENGTH_ONE_LANE if center_length is None else center_length
        return [(":bottom_0", junction_length), (":top_0", junction_length), (":right_0", junction_length),
                (":left_0", junction_length), (":center_0", c), (":center_1", c)]

    def specify_loop_order(self):
        """Edges in driving order around the closed route (internal edges included)."""
        return ["bottom", ":center_1", "top", ":top_0", "upper_ring", ":right_0", "right", ":center_0", "left",
                ":left_0", "lower_ring", ":bottom_0"]

    def specify_crossing(self):
        """(internal edge of the priority stream, internal edge of the yielding stream): edge priorities 78 vs
        46, figure_eight.py:126-154."""
        return ":center_1", ":center_0"
