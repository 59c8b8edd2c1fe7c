"""Pending-deprecation module (flow/scenarios/bottleneck.py): the old import path of flow/networks/bottleneck.py."""
from flow_amd.utils.flow_warnings import deprecated
from flow_amd.networks.bottleneck import BottleneckNetwork as _BottleneckNetwork
from flow_amd.networks.bottleneck import ADDITIONAL_NET_PARAMS  # noqa: F401


@deprecated('flow.scenarios.bottleneck', 'flow.networks.bottleneck.BottleneckNetwork')
class BottleneckScenario(_BottleneckNetwork):
    """See parent class."""

    pass
