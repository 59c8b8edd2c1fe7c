"""flow_amd: MI355X-native batched traffic micro-simulation behind Flow's Env API.

The simulation path is HIP-only (flow_amd/libflowsim.so, built from
flow_amd/csrc by ``python -m flow_amd.build``); importing the package does not
load the library, constructing a simulator does and fails loudly without it.
"""
__version__ = "0.1.0"
