"""Old package path flow.scenarios (now flow.networks), kept for experiment files written before the rename
(flow/scenarios/__init__.py).  Only the networks this package builds are here."""
from flow_amd.scenarios.base import Scenario
from flow_amd.scenarios.bottleneck import BottleneckScenario
from flow_amd.scenarios.figure_eight import Figure8Scenario, FigureEightScenario
from flow_amd.scenarios.loop import LoopScenario
from flow_amd.scenarios.merge import MergeScenario
from flow_amd.scenarios.multi_loop import MultiLoopScenario
from flow_amd.scenarios.multi_ring import MultiRingScenario
from flow_amd.scenarios.ring import RingScenario

__all__ = ["Scenario", "BottleneckScenario", "FigureEightScenario", "RingScenario", "MergeScenario",
           "MultiRingScenario", "Figure8Scenario", "LoopScenario", "MultiLoopScenario"]
