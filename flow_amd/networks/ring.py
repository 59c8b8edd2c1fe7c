"""RingNetwork (flow/networks/ring.py): four quarter-circle edges bottom -> right -> top -> left."""
from math import pi, sin, cos

import numpy as np

from flow_amd.core.params import InitialConfig, TrafficLightParams
from flow_amd.networks.base import Network

ADDITIONAL_NET_PARAMS = {
    # length of the ring road
    "length": 230,
    # number of lanes
    "lanes": 1,
    # speed limit for all edges
    "speed_limit": 30,
    # resolution of the curves on the ring
    "resolution": 40
}

FS_NETWORK = 0          # FS_NET_RING


class RingNetwork(Network):
    """flow/networks/ring.py:20-216."""

    def __init__(self, name, vehicles, net_params, initial_config=InitialConfig(),
                 traffic_lights=TrafficLightParams(), detector_params=None):
        for p in ADDITIONAL_NET_PARAMS.keys():
            if p not in net_params.additional_params:
                raise KeyError('Network parameter "{}" not supplied'.format(p))
        super().__init__(name, vehicles, net_params, initial_config, traffic_lights, detector_params)

    def specify_nodes(self, net_params):
        r = net_params.additional_params["length"] / (2 * pi)
        return [{"id": "bottom", "x": 0, "y": -r}, {"id": "right", "x": r, "y": 0},
                {"id": "top", "x": 0, "y": r}, {"id": "left", "x": -r, "y": 0}]

    def specify_edges(self, net_params):
        length = net_params.additional_params["length"]
        resolution = net_params.additional_params["resolution"]
        r = length / (2 * pi)
        edgelen = length / 4.
        order = [("bottom", "bottom", "right", -pi / 2), ("right", "right", "top", 0),
                 ("top", "top", "left", pi / 2), ("left", "left", "bottom", pi)]
        return [{"id": eid, "type": "edgeType", "from": frm, "to": to, "length": edgelen,
                 "shape": [(r * cos(t), r * sin(t)) for t in np.linspace(t0, t0 + pi / 2, resolution)]}
                for eid, frm, to, t0 in order]

    def specify_types(self, net_params):
        return [{"id": "edgeType", "numLanes": net_params.additional_params["lanes"],
                 "speed": net_params.additional_params["speed_limit"]}]

    def specify_routes(self, net_params):
        ring = ["bottom", "right", "top", "left"]
        return {e: ring[i:] + ring[:i] for i, e in enumerate(ring)}

    def specify_edge_starts(self):
        ring_length = self.net_params.additional_params["length"]
        junction_length = 0.1                      # networks/ring.py:197
        return [("bottom", 0), ("right", 0.25 * ring_length + junction_length),
                ("top", 0.5 * ring_length + 2 * junction_length),
                ("left", 0.75 * ring_length + 3 * junction_length)]

    def specify_internal_edge_starts(self):
        ring_length = self.net_params.additional_params["length"]
        junction_length = 0.1                      # networks/ring.py:209
        return [(":right_0", 0.25 * ring_length), (":top_0", 0.5 * ring_length + junction_length),
                (":left_0", 0.75 * ring_length + 2 * junction_length),
                (":bottom_0", ring_length + 3 * junction_length)]

    def specify_internal_edges(self, junction_length, center_length=None):
        return [(":right_0", junction_length), (":top_0", junction_length), (":left_0", junction_length),
                (":bottom_0", junction_length)]


def ring_start_positions(num_vehicles, length=230.0, bunching=0.0, min_gap=0.0, x0=0.0):
    """Absolute start positions of ``num_vehicles`` evenly spaced vehicles on a one-lane ring
    (what generate_starting_positions yields for RingNetwork; used by bench.py)."""
    from flow_amd.core.kernel.network import NetworkKernel
    from flow_amd.core.params import NetParams, VehicleParams
    net = RingNetwork("ring", VehicleParams(), NetParams(additional_params={
        "length": length, "lanes": 1, "speed_limit": 30, "resolution": 40}),
        InitialConfig(bunching=bunching, min_gap=min_gap, x0=x0))
    k = NetworkKernel(net)
    pos, _ = k.generate_starting_positions(net.initial_config, num_vehicles)
    return np.array([k.get_x(e, p) for e, p in pos], dtype=np.float64)
