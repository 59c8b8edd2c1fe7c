"""Networks built for the GPU hot path (names as in flow/networks/__init__.py)."""
from flow_amd.networks.base import Network
from flow_amd.networks.ring import RingNetwork
from flow_amd.networks.figure_eight import FigureEightNetwork
from flow_amd.networks.merge import MergeNetwork
from flow_amd.networks.bottleneck import BottleneckNetwork

__all__ = ["Network", "RingNetwork", "FigureEightNetwork", "MergeNetwork", "BottleneckNetwork"]
