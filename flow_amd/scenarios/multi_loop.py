"""Pending-deprecation module (flow/scenarios/multi_loop.py): the old import path of flow/networks/multi_ring.py."""
from flow_amd.utils.flow_warnings import deprecated
from flow_amd.networks.multi_ring import MultiRingNetwork as _MultiRingNetwork
from flow_amd.networks.multi_ring import ADDITIONAL_NET_PARAMS  # noqa: F401


@deprecated('flow.scenarios.multi_loop', 'flow.networks.multi_ring.MultiRingNetwork')
class MultiLoopScenario(_MultiRingNetwork):
    """See parent class."""

    pass
