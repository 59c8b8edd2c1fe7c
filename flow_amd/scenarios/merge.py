"""Pending-deprecation module (flow/scenarios/merge.py): the old import path of flow/networks/merge.py."""
from flow_amd.utils.flow_warnings import deprecated
from flow_amd.networks.merge import MergeNetwork as _MergeNetwork
from flow_amd.networks.merge import ADDITIONAL_NET_PARAMS  # noqa: F401


@deprecated('flow.scenarios.merge', 'flow.networks.merge.MergeNetwork')
class MergeScenario(_MergeNetwork):
    """See parent class."""

    pass
