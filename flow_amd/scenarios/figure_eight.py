"""Pending-deprecation module (flow/scenarios/figure_eight.py): the old import path of flow/networks/figure_eight.py."""
from flow_amd.utils.flow_warnings import deprecated
from flow_amd.networks.figure_eight import FigureEightNetwork as _FigureEightNetwork
from flow_amd.networks.figure_eight import FigureEightNetwork as _FigureEightNetwork
from flow_amd.networks.figure_eight import ADDITIONAL_NET_PARAMS  # noqa: F401


@deprecated('flow.scenarios.figure_eight', 'flow.networks.figure_eight.FigureEightNetwork')
class FigureEightScenario(_FigureEightNetwork):
    """See parent class."""

    pass


@deprecated('flow.scenarios.figure_eight', 'flow.networks.figure_eight.FigureEightNetwork')
class Figure8Scenario(_FigureEightNetwork):
    """See parent class."""

    pass
