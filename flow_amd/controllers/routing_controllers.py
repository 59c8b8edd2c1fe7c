"""Routing controller descriptors (flow/controllers/routing_controllers.py)."""


class BaseRouter:
    """flow/controllers/base_routing_controller.py:19-31."""

    def __init__(self, veh_id, router_params):
        self.veh_id = veh_id
        self.router_params = router_params


class ContinuousRouter(BaseRouter):
    """routing_controllers.py:8-42: keeps vehicles circulating on closed networks -- a closed
    loop is periodic by construction in the GPU simulator, so this is a tag only."""
