"""Pending-deprecation module (flow/scenarios/base.py): the old import path of flow/networks/base.py."""
from flow_amd.utils.flow_warnings import deprecated
from flow_amd.networks.base import Network as _Network


@deprecated('flow.scenarios.base', 'flow.networks.base.Network')
class Scenario(_Network):
    """See parent class."""

    pass
