"""MergeNetwork (flow/networks/merge.py): a one-lane highway with a single on-ramp.  An OPEN network:
vehicles enter through inflows on ``inflow_highway`` / ``inflow_merge`` and leave at the end of ``center``."""
from numpy import pi, sin, cos

from flow_amd.core.params import InitialConfig, TrafficLightParams
from flow_amd.networks.base import Network

INFLOW_EDGE_LEN = 100      # length of the inflow edges (needed for resets) -- merge.py:8
VEHICLE_LENGTH = 5

ADDITIONAL_NET_PARAMS = {
    # length of the merge edge
    "merge_length": 100,
    # length of the highway leading to the merge
    "pre_merge_length": 200,
    # length of the highway past the merge
    "post_merge_length": 100,
    # number of lanes in the merge
    "merge_lanes": 1,
    # number of lanes in the highway
    "highway_lanes": 1,
    # max speed limit of the network
    "speed_limit": 30,
}

# length Flow's own edge-start table assumes for the internal edges of node `center`
# (("center", INFLOW_EDGE_LEN + premerge + 22.6) after (":center", INFLOW_EDGE_LEN + premerge + 0.1), merge.py:201-216)
CENTER_JUNCTION_LENGTH = 22.5


class MergeNetwork(Network):
    """flow/networks/merge.py:27-218."""

    def __init__(self, name, vehicles, net_params, initial_config=InitialConfig(),
                 traffic_lights=TrafficLightParams(), detector_params=None):
        for p in ADDITIONAL_NET_PARAMS.keys():
            if p not in net_params.additional_params:
                raise KeyError('Network parameter "{}" not supplied'.format(p))
        super().__init__(name, vehicles, net_params, initial_config, traffic_lights, detector_params)

    def _lengths(self):
        ap = self.net_params.additional_params
        return ap["merge_length"], ap["pre_merge_length"], ap["post_merge_length"]

    def specify_nodes(self, net_params):
        ap = net_params.additional_params
        merge, pre, post = ap["merge_length"], ap["pre_merge_length"], ap["post_merge_length"]
        a = pi / 4
        return [{"id": "inflow_highway", "x": -INFLOW_EDGE_LEN, "y": 0}, {"id": "left", "x": 0, "y": 0},
                {"id": "center", "x": pre, "y": 0, "radius": 10}, {"id": "right", "x": pre + post, "y": 0},
                {"id": "inflow_merge", "x": pre - (merge + INFLOW_EDGE_LEN) * cos(a),
                 "y": -(merge + INFLOW_EDGE_LEN) * sin(a)},
                {"id": "bottom", "x": pre - merge * cos(a), "y": -merge * sin(a)}]

    def specify_edges(self, net_params):
        ap = net_params.additional_params
        rows = [("inflow_highway", "highwayType", "inflow_highway", "left", INFLOW_EDGE_LEN),
                ("left", "highwayType", "left", "center", ap["pre_merge_length"]),
                ("inflow_merge", "mergeType", "inflow_merge", "bottom", INFLOW_EDGE_LEN),
                ("bottom", "mergeType", "bottom", "center", ap["merge_length"]),
                ("center", "highwayType", "center", "right", ap["post_merge_length"])]
        return [{"id": i, "type": t, "from": f, "to": to, "length": ln} for (i, t, f, to, ln) in rows]

    def specify_types(self, net_params):
        ap = net_params.additional_params
        return [{"id": "highwayType", "numLanes": ap["highway_lanes"], "speed": ap["speed_limit"]},
                {"id": "mergeType", "numLanes": ap["merge_lanes"], "speed": ap["speed_limit"]}]

    def specify_routes(self, net_params):
        return {"inflow_highway": ["inflow_highway", "left", "center"], "left": ["left", "center"],
                "center": ["center"], "inflow_merge": ["inflow_merge", "bottom", "center"],
                "bottom": ["bottom", "center"]}

    def specify_edge_starts(self):
        _, pre, post = self._lengths()
        return [("inflow_highway", 0), ("left", INFLOW_EDGE_LEN + 0.1), ("center", INFLOW_EDGE_LEN + pre + 22.6),
                ("inflow_merge", INFLOW_EDGE_LEN + pre + post + 22.6),
                ("bottom", 2 * INFLOW_EDGE_LEN + pre + post + 22.7)]

    def specify_internal_edge_starts(self):
        _, pre, post = self._lengths()
        return [(":left", INFLOW_EDGE_LEN), (":center", INFLOW_EDGE_LEN + pre + 0.1),
                (":bottom", 2 * INFLOW_EDGE_LEN + pre + post + 22.6)]

    # ---- what netconvert would add (no netconvert here)
    def specify_internal_edges(self, junction_length, center_length=None):
        c = CENTER_JUNCTION_LENGTH if center_length is None else center_length
        return [(":left_0", junction_length), (":bottom_0", junction_length), (":center_0", c), (":center_1", c)]

    def specify_open_routes(self):
        """The two driving paths, internal edges included, major (priority) route first: the straight
        highway connection has the right of way over the turning on-ramp connection at `center`."""
        return [["inflow_highway", ":left_0", "left", ":center_1", "center"],
                ["inflow_merge", ":bottom_0", "bottom", ":center_0", "center"]]
