"""Host-side views that stand where flow/core/kernel/{network,vehicle,simulation} stood."""
from flow_amd.core.kernel.kernel import Kernel

__all__ = ["Kernel"]
