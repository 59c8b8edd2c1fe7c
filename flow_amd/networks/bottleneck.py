"""BottleneckNetwork (flow/networks/bottleneck.py): a straight road whose 4 lanes narrow to 2 and then to 1 at two
zipper junctions (the toll-plaza / lane-drop geometry of the bay-bridge experiments).  An OPEN network: vehicles
enter on edge "1" and leave at the end of edge "5"."""
import numpy as np

from flow_amd.core.params import InitialConfig, TrafficLightParams
from flow_amd.networks.base import Network

ADDITIONAL_NET_PARAMS = {
    # the factor multiplying number of lanes.
    "scaling": 1,
    # edge speed limit
    'speed_limit': 23
}

# edge id -> (length [m], lanes per unit of scaling) -- bottleneck.py:116-165
EDGES = (("1", 100, 4), ("2", 310, 4), ("3", 140, 4), ("4", 280, 2), ("5", 155, 1))
# netconvert is not run here: the junctions at the zipper nodes 4 and 5 (radius 20) get this internal length
ZIPPER_JUNCTION_LENGTH = 20.0


class BottleneckNetwork(Network):
    """flow/networks/bottleneck.py:17-240."""

    def __init__(self, name, vehicles, net_params, initial_config=InitialConfig(),
                 traffic_lights=TrafficLightParams(), detector_params=None):
        for p in ADDITIONAL_NET_PARAMS.keys():
            if p not in net_params.additional_params:
                raise KeyError('Network parameter "{}" not supplied'.format(p))
        super().__init__(name, vehicles, net_params, initial_config, traffic_lights, detector_params)

    def specify_nodes(self, net_params):
        return [{"id": "1", "x": 0, "y": 0}, {"id": "2", "x": 100, "y": 0}, {"id": "3", "x": 410, "y": 0},
                {"id": "4", "x": 550, "y": 0, "type": "zipper", "radius": 20},
                {"id": "5", "x": 830, "y": 0, "type": "zipper", "radius": 20}, {"id": "6", "x": 985, "y": 0},
                {"id": "fake1", "x": 0, "y": 1}, {"id": "fake2", "x": 0, "y": 2}]

    def specify_edges(self, net_params):
        scaling = net_params.additional_params.get("scaling", 1)
        speed = net_params.additional_params['speed_limit']
        assert isinstance(scaling, int), "Scaling must be an int"
        edges = [{"id": e, "from": e, "to": str(int(e) + 1), "length": ln, "spreadType": "center",
                  "numLanes": lanes * scaling, "speed": speed} for e, ln, lanes in EDGES]
        # the reference's extra edge, a rendering aid off the road (:165-174): nothing drives on it, but it is part of
        # get_edge_list() and so of BottleneckAccelEnv's observation and of the network length
        edges.append({"id": "fake_edge", "from": "fake1", "to": "fake2", "length": 1, "spreadType": "center",
                      "numLanes": scaling, "speed": speed})
        return edges

    def specify_connections(self, net_params):
        scaling = net_params.additional_params.get("scaling", 1)
        return {"4": [{"from": "3", "to": "4", "fromLane": i, "toLane": int(np.floor(i / 2))}
                      for i in range(4 * scaling)],
                "5": [{"from": "4", "to": "5", "fromLane": i, "toLane": int(np.floor(i / 2))}
                      for i in range(2 * scaling)]}

    def specify_centroids(self, net_params):
        return [{"id": "1", "from": None, "to": "1", "x": -30, "y": 0},
                {"id": "1", "from": "5", "to": None, "x": 985 + 30, "y": 0}]

    def specify_routes(self, net_params):
        names = [e for e, _, _ in EDGES]
        return {e: names[i:] for i, e in enumerate(names)}

    def specify_edge_starts(self):
        return [("1", 0), ("2", 100), ("3", 405), ("4", 425), ("5", 580)]

    def get_bottleneck_lanes(self, lane):
        """Return the reduced number of lanes."""
        return [int(lane / 2), int(lane / 4)]

    # ---- what netconvert would add (no netconvert here)
    def specify_internal_edges(self, junction_length, center_length=None):
        z = ZIPPER_JUNCTION_LENGTH if center_length is None else center_length
        return [(":2_0", junction_length), (":3_0", junction_length), (":4_0", z), (":5_0", z)]

    def specify_open_routes(self):
        """One driving path for every lane (internal edges included); the lanes differ only in where they join."""
        return [["1", ":2_0", "2", ":3_0", "3", ":4_0", "4", ":5_0", "5"]]

    def specify_lane_joins(self):
        """(edge at whose start lanes 2q and 2q+1 have become lane q) for the two zipper junctions."""
        return ["4", "5"]
