"""flow_amd: MI355X-native batched traffic micro-simulation behind Flow's Env API.

The simulation path is HIP-only (flow_amd/libflowsim.so, built from
flow_amd/csrc by ``python -m flow_amd.build``); importing the package does not
load the library, constructing a simulator does and fails loudly without it.
"""
__version__ = "0.1.0"


def install_as_flow():
    """Make ``import flow.xyz`` resolve to the SAME module objects as ``import flow_amd.xyz`` so that an
    experiment file written against the reference (``from flow.controllers import IDMController`` ...) runs
    unchanged.  Aliasing only ``sys.modules['flow']`` is not enough: sub-modules would be imported a second time
    under the ``flow.`` name and their classes would no longer be the ones this package checks against."""
    import importlib
    import importlib.abc
    import importlib.util
    import sys

    class _FlowAlias(importlib.abc.MetaPathFinder, importlib.abc.Loader):
        def find_spec(self, fullname, path=None, target=None):
            if fullname != "flow" and not fullname.startswith("flow."):
                return None
            try:
                importlib.import_module("flow_amd" + fullname[4:])
            except ImportError:
                return None
            return importlib.util.spec_from_loader(fullname, self)

        def create_module(self, spec):
            return sys.modules["flow_amd" + spec.name[4:]]

        def exec_module(self, module):
            pass

    if not any(type(f).__name__ == "_FlowAlias" for f in sys.meta_path):
        sys.meta_path.insert(0, _FlowAlias())
    sys.modules["flow"] = sys.modules[__name__]
