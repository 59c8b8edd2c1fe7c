"""Pending-deprecation module (flow/scenarios/ring.py): the old import path of flow/networks/ring.py."""
from flow_amd.utils.flow_warnings import deprecated
from flow_amd.networks.ring import RingNetwork as _RingNetwork
from flow_amd.networks.ring import ADDITIONAL_NET_PARAMS  # noqa: F401


@deprecated('flow.scenarios.ring', 'flow.networks.ring.RingNetwork')
class RingScenario(_RingNetwork):
    """See parent class."""

    pass
