"""CPU-side checks of the drop-in boundary: libflowsim.so loads, exports every symbol
include/flowsim.h declares, and the ctypes mirror has the C layout (no compute calls)."""
import ctypes
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "flowsim.h")


@pytest.fixture(scope="module")
def lib():
    from flow_amd import build
    build.build()
    from flow_amd import _lib
    return _lib.load()


def declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(fs_[a-z_0-9]+)\s*\(", text)))


def test_header_declares_the_expected_entry_points():
    from flow_amd import _lib
    assert declared_symbols() == sorted(_lib.EXPORTS)


def test_library_exports_every_declared_symbol(lib):
    for name in declared_symbols():
        assert hasattr(lib, name), name
    from flow_amd import _lib
    assert lib.fs_abi_version() == _lib.FS_ABI_VERSION == 8


def test_ctypes_layout_matches_the_c_header(tmp_path):
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "flowsim.h"\n'
                   'int main(void){printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\\n", sizeof(fs_config), sizeof(fs_vehicle_spec),'
                   ' offsetof(fs_config, seed), offsetof(fs_config, vehicles), offsetof(fs_vehicle_spec, noise),'
                   ' offsetof(fs_config, junction), sizeof(fs_segment), sizeof(fs_inflow), offsetof(fs_config, inflows),'
                   ' offsetof(fs_config, route_start), offsetof(fs_config, ma_apply_actions));'
                   'return 0;}\n')
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    out = subprocess.check_output([str(exe)]).decode().split()
    from flow_amd import _lib
    assert int(out[0]) == ctypes.sizeof(_lib.fs_config)
    assert int(out[1]) == ctypes.sizeof(_lib.fs_vehicle_spec)
    assert int(out[2]) == _lib.fs_config.seed.offset
    assert int(out[3]) == _lib.fs_config.vehicles.offset
    assert int(out[4]) == _lib.fs_vehicle_spec.noise.offset
    assert int(out[5]) == _lib.fs_config.junction.offset and int(out[6]) == ctypes.sizeof(_lib.fs_segment)
    assert int(out[7]) == ctypes.sizeof(_lib.fs_inflow) and int(out[8]) == _lib.fs_config.inflows.offset
    assert int(out[9]) == _lib.fs_config.route_start.offset and int(out[10]) == _lib.fs_config.ma_apply_actions.offset


def test_enums_agree_between_header_binding_and_oracle():
    from flow_amd import _lib
    from oracle import refsim as S
    text = open(HEADER).read()

    def enum_val(name):
        m = re.search(r"\b%s\s*=\s*(-?\d+)" % name, text)
        assert m, name
        return int(m.group(1))
    for n in ("SIM", "RL", "IDM", "CFM", "BCM", "LAC", "OVM", "LINEAR_OVM", "GIPPS", "FOLLOWER_STOPPER",
              "NONLOCAL_FOLLOWER_STOPPER", "PISATURATION"):
        assert enum_val("FS_CTRL_" + n) == getattr(_lib, "FS_CTRL_" + n) == getattr(S, "CTRL_" + n)
    for n in ("ACCEL", "WAVE_ATTENUATION", "WAVE_ATTENUATION_PO", "LANE_CHANGE_ACCEL"):
        assert enum_val("FS_ENV_" + n) == getattr(_lib, "FS_ENV_" + n) == getattr(S, "ENV_" + n)
    for n in ("NONE", "INSTANTANEOUS", "SAFE_VELOCITY"):
        assert enum_val("FS_FAILSAFE_" + n) == getattr(_lib, "FS_FAILSAFE_" + n) == getattr(S, "FAILSAFE_" + n)


def test_create_fails_loudly_without_a_gpu(lib):
    """No CPU fallback: on a box without a HIP device fs_create must return an error."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import ring_spec
    from flow_amd.sim import FlowSim
    from flow_amd.utils.exceptions import FatalFlowError
    with pytest.raises(FatalFlowError):
        FlowSim(ring_spec(R=1, N=5, bunching=0), "f32")


def test_config_validation_needs_no_gpu(lib):
    """fs_create validates the whole config before it touches the device: the reference's error types come
    back through the binding (ValueError / NotImplementedError / FatalFlowError) with a message."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import idm_vehicle, ring_spec
    from flow_amd.sim import FlowSim
    from flow_amd.utils.exceptions import FatalFlowError

    def spec(**kw):
        s = ring_spec(R=2, N=5, bunching=0)
        s.update(kw)
        return s
    with pytest.raises(ValueError, match="unknown controller"):
        FlowSim(spec(vehicles=[idm_vehicle(controller=99)] * 5), "f32")
    with pytest.raises(ValueError, match="rl_index"):
        FlowSim(spec(vehicles=[idm_vehicle(controller=1, rl_index=3)] * 5, num_rl=5), "f32")      # used twice
    with pytest.raises(ValueError, match="rl_index out of range"):
        FlowSim(spec(vehicles=[idm_vehicle()] * 4 + [idm_vehicle(controller=1, rl_index=4)], num_rl=1), "f32")
    with pytest.raises(ValueError, match="num_rl"):
        FlowSim(spec(num_rl=2), "f32")
    with pytest.raises(ValueError, match="sim_step"):
        FlowSim(spec(sim_step=0.0), "f32")
    with pytest.raises(ValueError, match="slowdown_ramp"):
        FlowSim(spec(slowdown_ramp=1.5), "f32")
    with pytest.raises(ValueError, match="init_pos"):
        FlowSim(spec(init_pos=np.full((2, 5), 500.0)), "f32")
    with pytest.raises(FatalFlowError, match="do not fit"):              # network/base.py:603-605
        FlowSim(spec(ring_length=np.full(2, 20.0), init_pos=np.tile(np.arange(5) * 3.0, (2, 1))), "f32")
    big = ring_spec(R=1, N=65, length=800.0, bunching=0)
    with pytest.raises(NotImplementedError, match="64 vehicles"):
        FlowSim(big, "f32")
    with pytest.raises(NotImplementedError, match="multi-lane"):
        FlowSim(spec(num_lanes=2, env=2, num_rl=1,
                     vehicles=[idm_vehicle()] * 4 + [idm_vehicle(controller=1, rl_index=0)]), "f32")
    with pytest.raises(ValueError, match="init_lane"):
        FlowSim(spec(num_lanes=2, init_lane=np.full((2, 5), 7)), "f32")
    with pytest.raises(ValueError, match="segment table"):
        FlowSim(dict(spec(), junction=dict(a_in=1, a_out=2, b_in=3, b_out=4, lookahead=1, time_gap=1, za_lo=0,
                                           za_hi=1, zb_lo=0, zb_hi=1)), "f32")


def test_integration_doc_binding_lists_the_current_fields():
    """INTEGRATION.md shows the ctypes stub a maintainer of the reference would add: its field lists must be the
    binding's (they are regenerated from flow_amd/_lib.py whenever the ABI changes)."""
    from flow_amd import _lib
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    for cls in (_lib.fs_vehicle_spec, _lib.fs_segment, _lib.fs_inflow, _lib.fs_cell, _lib.fs_junction, _lib.fs_config):
        m = re.search(r"class %s\(C\.Structure\):\n(.*?)\]\n" % cls.__name__, text, flags=re.S)
        assert m, cls.__name__
        assert re.findall(r'\("([a-z_0-9]+)",', m.group(1)) == [n for n, _ in cls._fields_], cls.__name__
