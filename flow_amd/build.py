"""Build libflowsim.so (HIP, gfx950) in-tree.  hipcc cross-compiles without a GPU."""
import os
import shutil
import subprocess

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
SRC = os.path.join(PKG, "csrc", "flowsim.hip")
DEPS = [SRC, os.path.join(PKG, "csrc", "flowsim_kernels.h"), os.path.join(PKG, "csrc", "flowsim_open.h"),
        os.path.join(PKG, "csrc", "flowsim_wide.h"), os.path.join(PKG, "csrc", "flowsim_pair.h"), os.path.join(PKG, "csrc", "flowsim_pair_step_a.inc"),
        os.path.join(PKG, "csrc", "flowsim_pair_step_a_sm.inc"), os.path.join(PKG, "csrc", "flowsim_fig8.h"),
        os.path.join(ROOT, "include", "flowsim.h")]
LIB = os.path.join(PKG, "libflowsim.so")

# -ffp-contract=off: the kernels are the float32 bit-twin of the oracle only if a*b+c is never fused
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fPIC", "-shared", "-std=c++17",
               "-Wall", "-Wno-unused-function"]


def find_hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: libflowsim.so cannot be built (no CPU fallback exists)")


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(d) > t for d in DEPS)


def build(force=False, verbose=False):
    """Compile flow_amd/csrc/flowsim.hip -> flow_amd/libflowsim.so; returns the path."""
    if not force and not needs_build():
        return LIB
    cmd = [find_hipcc()] + HIPCC_FLAGS + ["-I" + os.path.join(ROOT, "include"),
                                          "-I" + os.path.join(PKG, "csrc"), "-o", LIB + ".tmp", SRC]
    if verbose:
        print(" ".join(cmd))
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + res.stdout + res.stderr)
    os.replace(LIB + ".tmp", LIB)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
