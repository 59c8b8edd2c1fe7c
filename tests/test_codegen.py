"""Code-generation guards (no GPU needed: hipcc cross-compiles gfx950 here).

The step kernels read per-replica values other lanes hold with v_readlane (segment tables of the closed loops,
ordering keys, inflow counters).  That idiom is only sound while those VGPRs are never parked in AGPRs: hipcc
re-materialises an AGPR-held value with v_accvgpr_read under the CURRENT exec mask right before the v_readlane, so
rows held by lanes that are inactive at that point would read stale data (found in the float64 wide kernel, docs/HISTORY.md
section 4; the open-network kernels read their launch tables from LDS for that reason).  This test pins the premise:
every float32 instantiation of the step kernels uses zero AGPRs and spills nothing to scratch memory -- it caught
k_steps_open<float, 32, .> growing past 256 VGPRs once."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def device_asm(tmp_path_factory):
    """Device assembly of every object of the library (flow_amd/build.py: parts()), concatenated."""
    from flow_amd import build
    try:
        build.find_hipcc()
    except RuntimeError:
        pytest.skip("hipcc not found")
    files = build.device_asm(str(tmp_path_factory.mktemp("asm")))
    return "\n".join(open(f).read() for f in files.values())


def kernel_resources(asm):
    """{mangled kernel name: {num_vgpr, num_agpr, private_seg_size}} from the .set directives of the device asm."""
    table = {}
    for name, key, val in re.findall(r"\.set (_ZN2fs\w+)\.(num_vgpr|num_agpr|private_seg_size), (\d+)", asm):
        table.setdefault(name, {})[key] = int(val)
    return table


def test_float32_step_kernels_use_no_agprs(device_asm):
    table = kernel_resources(device_asm)
    step_kernels = {n: r for n, r in table.items()
                    if re.match(r"_ZN2fs\d+(k_steps|k_steps_ml|k_steps_open|k_steps_wide|k_rollout_idm|k_rollout_pair|"
                                r"k_rollout_loop)I", n)}
    f32 = {n: r for n, r in step_kernels.items() if re.match(r"_ZN2fs\d+k_\w+?If", n)}
    f64 = {n: r for n, r in step_kernels.items() if re.match(r"_ZN2fs\d+k_\w+?Id", n)}
    assert len(f32) >= 20 and len(f64) >= 10, (len(f32), len(f64))
    # (a few bytes of scratch are fine: the large-argument path of libm's cosf keeps a small array there)
    bad = {n: r for n, r in f32.items() if r.get("num_agpr", 0) != 0 or r.get("private_seg_size", 0) > 64}
    assert not bad, bad
    # the headline kernel must leave room for 2+ waves per SIMD (512 VGPRs per SIMD lane)
    rollout = [r for n, r in f32.items() if "k_rollout_idm" in n]
    assert rollout and max(r["num_vgpr"] for r in rollout) <= 128
    # float64 kernels that need more than 256 registers keep the surplus in AGPRs.  That is only safe where no
    # lane-held TABLE is read with v_readlane (a value re-materialised from an AGPR under a partial exec mask right
    # before the v_readlane was the wide kernel's miscompile): every table of the step kernels lives in LDS now
    # (open / wide: OpenTabs; closed loops: SegTab), so the generic kernels may appear here -- nothing else
    spilling = sorted({re.match(r"_ZN2fs\d+(k_[a-z_0-9]+?)I", n).group(1) for n, r in f64.items() if r.get("num_agpr", 0)})
    assert set(spilling) <= {"k_steps", "k_steps_open", "k_steps_wide"}, spilling
    # the two-vehicles-per-lane kernels pin v112..v145 by hand: they must stay well inside the VGPR file
    pair = [r for n, r in table.items() if "k_rollout_pair" in n]
    # (the FS_MIXED instantiation with speed-mode clamps is the largest: float64 state + two controllers' constants)
    assert pair and max(r["num_vgpr"] for r in pair) <= 192 and all(r.get("num_agpr", 0) == 0 for r in pair)


def test_hand_written_pair_step_keeps_its_registers(device_asm):
    """flowsim_pair.h pins v112..v147 by name inside its asm blocks (flowsim_pair_step_a*.inc): the values that live
    across the blocks (speeds, positions, headways, v - v_leader: v[112:119]) are operands bound to those registers.
    The assignment was validated on ROCm VALIDATED_ROCM; a compiler that stopped honouring the pins, or that started to
    park something of its own in them between the blocks, would corrupt the step silently on the CPU box (the GPU
    parity tests would catch it a round late).  Checked on the disassembly of the hot float32 instantiation:
      * the toolchain is the validated release;
      * between the asm blocks of the unrolled 16-step body the compiler's own instructions never WRITE v112..v119 and
        never read a temporary both blocks clobber (v124..v127, v130..v133) that it has not written itself since the
        last block (the blocks' outputs -- v[128:129], the `=&v` operand of part A -- are the compiler's to read);
      * no vector-memory LOAD appears inside that body (nothing asynchronous can feed a block)."""
    from flow_amd import build
    ver = subprocess.run([build.find_hipcc(), "--version"], capture_output=True, text=True).stdout
    m = re.search(r"HIP version: (\d+\.\d+)", ver)
    assert m and m.group(1) == build.VALIDATED_ROCM, "the hand-written register assignment was validated on ROCm %s: " \
        "re-validate (tests/test_pair_gpu.py on a GPU) and bump flow_amd/build.py VALIDATED_ROCM" % build.VALIDATED_ROCM
    name = re.search(r"^(_ZN2fs14k_rollout_pairIfLi16ELb1ELb1ELb0ELb0ELb0EE\S+?):", device_asm, re.M).group(1)
    body = device_asm[device_asm.index("\n" + name + ":"):]
    body = body[:body.index(".set " + name)]
    first, last = body.index(";;#ASMSTART"), body.rindex(";;#ASMEND")
    assert body.count(";;#ASMSTART") == body.count(";;#ASMEND") >= 32          # 16 unrolled steps x (part A, part B)
    outside = re.split(r";;#ASMSTART.*?;;#ASMEND", body[first:last + len(";;#ASMEND")], flags=re.S)

    def regs(tok):
        out = []
        for a, b in re.findall(r"\bv\[(\d+):(\d+)\]", tok):
            out += list(range(int(a), int(b) + 1))
        out += [int(a) for a in re.findall(r"\bv(\d+)\b", tok)]
        return out

    for seg in outside:
        written = set()
        for line in seg.split("\n"):
            t = line.strip()
            if not t or t.startswith(";") or t.startswith(".") or t.endswith(":") or t.startswith("s_"):
                continue
            assert not t.startswith(("global_load", "buffer_load", "flat_load", "scratch_load")), t
            op, _, rest = t.partition(" ")
            args = rest.split(",")
            is_store = op.startswith(("buffer_store", "global_store", "ds_write"))
            dst = [] if is_store else regs(args[0])
            src = regs(rest) if is_store else regs(",".join(args[1:]))
            assert not any(112 <= r <= 119 for r in dst), "compiler code writes a pinned state register: " + t
            stale = [r for r in src if (124 <= r <= 127 or 130 <= r <= 133) and r not in written]
            assert not stale, "compiler code reads a clobbered temporary of the asm blocks: " + t
            written.update(dst)
