"""Build libflowsim.so (HIP, gfx950) in-tree.  hipcc cross-compiles without a GPU.

The library is several objects: `flowsim.hip` (C ABI, validation, handle) and `flowsim_part.hip` compiled once per
(precision, lanes per replica) pair -- each pair instantiates its own step kernels, so the objects compile in parallel.
Objects are cached under flow_amd/csrc/_obj/ with a stamp = hash of the CONTENTS of every source and header plus the
flags: an edit to any of them recompiles every object (they share the headers' struct layouts), a change of flags only
the objects it concerns.  The library is current only if every object's stamp is (needs_build)."""
import concurrent.futures
import hashlib
import os
import shutil
import subprocess

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
SRC = os.path.join(CSRC, "flowsim.hip")
PART = os.path.join(CSRC, "flowsim_part.hip")
OBJ = os.path.join(CSRC, "_obj")
LIB = os.path.join(PKG, "libflowsim.so")

# the ROCm release the hand-written register assignment of k_rollout_pair (v112..v145, flowsim_pair_step_a*.inc) was
# validated on; tests/test_codegen.py re-checks the assignment against the disassembly on every build
VALIDATED_ROCM = "7.2"

HEADERS = ["flowsim_sim.h", "flowsim_launch.h", "flowsim_kernels.h", "flowsim_open.h", "flowsim_wide.h",
           "flowsim_pair.h", "flowsim_pair_step_a.inc", "flowsim_pair_step_a_sm.inc", "flowsim_fig8.h", "flowsim_ringrl.h", "flowsim_policy.h", "flowsim_queue.h", "flowsim_queue_consts.h", "flowsim_dropq.h",
           "flowsim_part.hip"]
DEPS = [SRC] + [os.path.join(CSRC, h) for h in HEADERS] + [os.path.join(ROOT, "include", "flowsim.h")]
assert all(os.path.exists(d) for d in DEPS)
_known = set(os.path.basename(d) for d in DEPS)
assert all(f in _known for f in os.listdir(CSRC) if f.endswith((".h", ".inc", ".hip"))), "flow_amd/build.py: a source under csrc/ is not in DEPS"

# -ffp-contract=off: the kernels are the float32 bit-twin of the oracle only if a*b+c is never fused
COMMON_FLAGS = ["--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-std=c++17", "-Wall", "-Wno-unused-function"]
HIPCC_FLAGS = COMMON_FLAGS + ["-fPIC", "-shared"]       # (kept for callers that compile a single file themselves)


def parts():
    """[(object name, source, extra -D flags)] of the library."""
    out = [("main", SRC, [])]
    for t, tn in (("float", "f32"), ("double", "f64")):
        for seg in (8, 16, 32, 64):
            out.append(("seg%d_%s" % (seg, tn), PART, ["-DFS_PART_T=" + t, "-DFS_PART_SEG=%d" % seg]))
        for w in (2, 4):
            out.append(("wide%d_%s" % (w, tn), PART, ["-DFS_PART_T=" + t, "-DFS_PART_WIDE=%d" % w]))
    out.append(("queue_f32", PART, ["-DFS_PART_T=float", "-DFS_PART_QUEUE=1"] +
                (["-DFS_QDIAG=1"] if os.environ.get("FLOWSIM_QDIAG") else [])))     # cycle counters of k_merge_queue's sections
    return out


def find_hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: libflowsim.so cannot be built (no CPU fallback exists)")


def include_flags():
    return ["-I" + os.path.join(ROOT, "include"), "-I" + CSRC]


def _stamp(extra):
    h = hashlib.sha256()
    for d in DEPS:
        h.update(os.path.relpath(d, ROOT).encode())     # (relative: the same tree at another path has the same stamps)
        with open(d, "rb") as f:
            h.update(hashlib.sha256(f.read()).digest())
    h.update(" ".join(COMMON_FLAGS + extra).encode())
    return h.hexdigest()[:16]


def _have_stamp(out):
    if not (os.path.exists(out) and os.path.exists(out + ".stamp")):
        return None
    with open(out + ".stamp") as f:
        return f.read().strip()


def needs_build():
    """The library is missing, older than a source, or some object was built from other sources / flags than the current
    ones (build(only=...) links stale objects: their stamps say so and the next plain build() recompiles them)."""
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    if any(os.path.getmtime(d) > t for d in DEPS):
        return True
    return any(_have_stamp(os.path.join(OBJ, name + ".o")) != _stamp(extra) for name, _, extra in parts())


def _compile(job):
    name, src, extra, out, verbose = job
    cmd = [find_hipcc()] + COMMON_FLAGS + ["-fPIC", "-c"] + extra + include_flags() + ["-o", out + ".tmp", src]
    if verbose:
        print(" ".join(cmd), flush=True)
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("hipcc failed (%s):\n%s%s" % (name, res.stdout, res.stderr))
    os.replace(out + ".tmp", out)
    return out


def _jobs():
    return max(1, min(len(parts()), int(os.environ.get("FLOWSIM_BUILD_JOBS", os.cpu_count() or 4))))


def build_variant(lib, defines, names=None, verbose=False):
    """A development build of the library with extra -D flags (phase timers, diagnostics) at path `lib`; objects under
    _obj/<basename>/.  `names`: the parts to compile with the flags (the others are taken from the plain build)."""
    build()
    obj_dir = os.path.join(OBJ, os.path.basename(lib))
    os.makedirs(obj_dir, exist_ok=True)
    objs, todo = [], []
    for name, src, extra in parts():
        if names and name not in names:
            objs.append(os.path.join(OBJ, name + ".o"))
            continue
        out = os.path.join(obj_dir, name + ".o")
        objs.append(out)
        todo.append((name, src, extra + list(defines), out, verbose))
    with concurrent.futures.ThreadPoolExecutor(_jobs()) as pool:
        list(pool.map(_compile, todo))
    res = subprocess.run([find_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs,
                         capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("hipcc link failed:\n" + res.stdout + res.stderr)
    return lib


def build(force=False, verbose=False, only=None):
    """Compile the objects (in parallel) and link flow_amd/libflowsim.so; returns its path.  `only` = names of the
    parts to recompile whatever their stamps say (development)."""
    if not force and not only and not needs_build():
        return LIB
    os.makedirs(OBJ, exist_ok=True)
    objs, todo, stamps = [], [], {}
    for name, src, extra in parts():
        out = os.path.join(OBJ, name + ".o")
        stamp, have = _stamp(extra), _have_stamp(out)
        objs.append(out)
        # development (`only`): the named parts are recompiled, every other object is linked as it is -- its stale
        # stamp makes needs_build() true, so the next plain build() recompiles it
        if force or (only and name in only) or (not only and have != stamp) or not os.path.exists(out):
            todo.append((name, src, extra, out, verbose))
            stamps[out] = stamp
    with concurrent.futures.ThreadPoolExecutor(_jobs()) as pool:
        for out in pool.map(_compile, todo):
            with open(out + ".stamp", "w") as f:
                f.write(stamps[out])
    cmd = [find_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB + ".tmp"] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("hipcc link failed:\n" + res.stdout + res.stderr)
    os.replace(LIB + ".tmp", LIB)
    return LIB


USER_DIR = os.path.join(PKG, "_user")
USER_SIGNATURE = ("template <typename T>\n__device__ __forceinline__ T fs_user_accel(T v, T v_lead, T h, bool has_lead, "
                  "T v_follow, T h_follow, T dt, T max_accel, const T* p)")


def user_header(body):
    """The header a user controller is compiled from: ``body`` = the statements of its get_accel (C++, ending in a
    return), with the vehicle's speed ``v``, the leader's ``v_lead``, the bumper-to-bumper headway ``h``, ``has_lead``,
    the follower's speed and headway, the step ``dt``, the vehicle type's ``max_accel`` and its parameters ``p[0..7]`` in
    scope; T is float or double (the handle's precision).  flowsim_kernels.h includes it inside namespace fs, after the
    helper functions (tmin / tmax / tabs / tsqrt)."""
    body = "\n".join("  " + ln.strip() for ln in str(body).strip().splitlines())    # (the text is the cache key: normalised)
    return ("// generated by flow_amd.build.build_user -- the get_accel of a flow_amd.controllers.CompiledController\n"
            "%s {\n%s\n}\n" % (USER_SIGNATURE, body))


def build_user(body, verbose=False):
    """A copy of the library whose FS_CTRL_USER slots run the user's controller: flow_amd/_user/<hash>/libflowsim.so.
    Every part that holds generic step kernels is recompiled with -DFS_USER_CONTROLLER_HEADER (in parallel; the stock
    objects of the C ABI and the queue kernels are reused); cached by the hash of the body and of the sources."""
    text = user_header(body)
    tag = hashlib.sha256((text + _stamp([])).encode()).hexdigest()[:16]
    out_dir = os.path.join(USER_DIR, tag)
    lib = os.path.join(out_dir, "libflowsim.so")
    if os.path.exists(lib):
        return lib
    build()
    if not all(os.path.exists(os.path.join(OBJ, name + ".o")) for name, _, _ in parts()):
        build(force=True)                              # (a tree that holds the library but not its objects)
    os.makedirs(out_dir, exist_ok=True)
    for stale in os.listdir(USER_DIR):                 # copies built from other sources than the current ones
        tagfile = os.path.join(USER_DIR, stale, "sources.stamp")
        if stale != tag and os.path.exists(tagfile) and open(tagfile).read().strip() != _stamp([]):
            shutil.rmtree(os.path.join(USER_DIR, stale), ignore_errors=True)
    with open(os.path.join(out_dir, "sources.stamp"), "w") as f:
        f.write(_stamp([]))
    hdr = os.path.join(out_dir, "user_controller.h")
    with open(hdr, "w") as f:
        f.write(text)
    objs, todo = [], []
    for name, src, extra in parts():
        if name == "main" or name.startswith("queue"):
            objs.append(os.path.join(OBJ, name + ".o"))
            continue
        out = os.path.join(out_dir, name + ".o")
        objs.append(out)
        todo.append((name, src, extra + ['-DFS_USER_CONTROLLER_HEADER="%s"' % hdr], out, verbose))
    with concurrent.futures.ThreadPoolExecutor(_jobs()) as pool:
        list(pool.map(_compile, todo))
    res = subprocess.run([find_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib + ".tmp"] + objs,
                         capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("hipcc link failed:\n" + res.stdout + res.stderr)
    os.replace(lib + ".tmp", lib)
    for job in todo:                                   # (the objects are not needed again: the library is the cache)
        os.remove(job[3])
    return lib


def device_asm(out_dir, names=None):
    """Device assembly (gfx950) of the parts, for the code-generation guards: {part name: path of its .s}."""
    os.makedirs(out_dir, exist_ok=True)
    jobs = []
    for name, src, extra in parts():
        if names and name not in names:
            continue
        out = os.path.join(out_dir, name + ".s")
        jobs.append((name, [find_hipcc()] + COMMON_FLAGS + ["-S", "--cuda-device-only"] + extra + include_flags() +
                     ["-o", out, src], out))

    def run(job):
        res = subprocess.run(job[1], capture_output=True, text=True)
        if res.returncode != 0:
            raise RuntimeError("hipcc -S failed (%s):\n%s" % (job[0], res.stderr[-2000:]))
        return job[0], job[2]

    with concurrent.futures.ThreadPoolExecutor(_jobs()) as pool:
        return dict(pool.map(run, jobs))


if __name__ == "__main__":
    import sys
    print(build(force="--force" in sys.argv, verbose=True, only=[a for a in sys.argv[1:] if not a.startswith("-")]))
