"""Trajectory ("emission") output in the CSV layout of flow/core/util.py:36-99.

The reference lets SUMO write an emission XML every simulation step and converts it with
``emission_to_csv``; downstream tools (flow/visualize/time_space_diagram.py:57-66) read the CSV
columns ``time, id, edge_id, relative_position, speed, lane_number`` (+ ``x, y``).  There is no
SUMO here, so the environment records those columns itself and writes the CSV directly.
SUMO's pollutant columns (CO, CO2, NOx, fuel, ...) are not produced.
"""
import csv
import errno
import math
import os

COLUMNS = ["time", "id", "x", "y", "angle", "type", "route", "speed", "edge_id", "lane_number",
           "relative_position"]


def ensure_dir(path):
    """flow/core/util.py:102-110."""
    try:
        os.makedirs(path)
    except OSError as exception:
        if exception.errno != errno.EEXIST:
            raise
    return path


class TrajectoryRecorder(object):
    """Collects one row per vehicle per recorded step for one environment (replica 0)."""

    def __init__(self, env):
        self.env = env
        self.rows = []

    def _xy(self, s):
        """Cartesian position on a ring laid out as in flow/networks/ring.py:70-168 (counter-clockwise
        from the bottom node); other networks report the loop coordinate on x."""
        add = self.env.net_params.additional_params
        if "length" in add:
            L = self.env.k.network.length()
            r = add["length"] / (2 * math.pi)
            th = -math.pi / 2 + 2 * math.pi * s / L
            return r * math.cos(th), r * math.sin(th), (math.degrees(-th) % 360)
        return s, 0.0, 0.0

    def record(self, time_s):
        veh = self.env.k.vehicle
        for vid in veh.get_ids():
            s = veh.get_x_by_id(vid)
            edge, rel = veh.get_edge(vid), veh.get_position(vid)
            x, y, ang = self._xy(s)
            self.rows.append({"time": round(time_s, 6), "id": vid, "x": x, "y": y, "angle": ang,
                              "type": veh.get_type(vid), "route": "route" + str(edge).lstrip(":").split("_")[0],
                              "speed": veh.get_speed(vid), "edge_id": edge, "lane_number": veh.get_lane(vid),
                              "relative_position": rel})

    def write(self, path):
        rows = sorted(self.rows, key=lambda k: k['id'])      # flow/core/util.py:87 (stable: time order kept)
        with open(path, 'w', newline='') as f:
            w = csv.DictWriter(f, COLUMNS)
            w.writeheader()
            w.writerows(rows)
        return path
