"""Build `libsed_hip.so` (the C-ABI library declared in include/sed_hip.h) for gfx950 with hipcc.

Cross-compiles without a GPU.  The .so is built IN-TREE (next to this file) so it travels with the repo
snapshot to the GPU box; it is git-ignored.
"""
import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libsed_hip.so")
SOURCES = ["logmel.hip", "bn.hip", "conv.hip", "conv_wino2.hip", "conv_sf16.hip", "gemm_sf16.hip", "heads.hip", "attention.hip", "gru.hip"]
INCLUDE = os.path.join(os.path.dirname(HERE), "include")
BASE_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-I", INCLUDE]
# SED_HIPCC_FLAGS: extra flags for kernel experiments (tools/ablate.sh, tools/experiments/*.patch).  A library built with them
# carries a different flags hash in sed_version(), and _lib.lib() refuses it unless SED_ALLOW_EXPERIMENT=1.
FLAGS = BASE_FLAGS + os.environ.get("SED_HIPCC_FLAGS", "").split()


def flags_hash(flags=None):
    """Identity of a set of hipcc flags (the include PATH is machine-specific and left out)."""
    flags = list(BASE_FLAGS if flags is None else flags)
    keep, skip = [], False
    for f in flags:
        if skip:
            skip = False
            continue
        if f == "-I":
            skip = True
            continue
        if f.startswith("-DSED_BUILD_FLAGS_HASH"):
            continue
        keep.append(f)
    return hashlib.sha1(" ".join(keep).encode()).hexdigest()[:12]


def source_hashes():
    """{file name: sha1[:12]} of every kernel source (csrc/*.hip, csrc/common.h).  Profile digests under profiles/ record it
    (tools/pmc_digest.py `_meta`), and bench.py quotes a committed PMC figure only while the sources of THAT kernel are the
    ones it was collected on."""
    out = {}
    for name in sorted(SOURCES + ["common.h"]):
        with open(os.path.join(CSRC, name), "rb") as f:
            out[name] = hashlib.sha1(f.read()).hexdigest()[:12]
    return out


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def compile_cmd(src, obj, flags=None):
    flags = list(FLAGS if flags is None else flags)
    return [os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")] + flags + ['-DSED_BUILD_FLAGS_HASH="%s"' % flags_hash(flags), "-c", src, "-o", obj]


def build(force=False, verbose=False):
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    hdrs = [os.path.join(CSRC, "common.h"), os.path.join(INCLUDE, "sed_hip.h")]
    # objects compiled with other flags (an experiment build) are never mixed with these: the stamp forces a full rebuild
    stamp = os.path.join(objdir, "flags.stamp")
    want = flags_hash(FLAGS)
    if not os.path.exists(stamp) or open(stamp).read().strip() != want:
        force = True
    jobs = []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(objdir, s.replace(".hip", ".o"))
        if force or _stale(obj, [src] + hdrs):
            jobs.append(compile_cmd(src, obj))

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n%s\n%s" % (" ".join(cmd), r.stderr))
        return r

    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(run, jobs))
    with open(stamp, "w") as f:
        f.write(want)
    objs = [os.path.join(objdir, s.replace(".hip", ".o")) for s in SOURCES]
    if force or jobs or _stale(LIB, objs):
        # Link with the host C++ driver and NO DT_NEEDED on libamdhip64: the hip* symbols (and the fat-binary
        # registration) must bind to the ONE HIP runtime already loaded in the process -- PyTorch-ROCm bundles its own
        # libamdhip64.so (soname without version), and a second runtime from /opt/rocm would not know torch's streams
        # and allocations.  _lib.py loads torch's runtime RTLD_GLOBAL first; a plain-C consumer links libamdhip64 itself.
        run([os.environ.get("CXX", "g++"), "-shared", "-fPIC", "-o", LIB] + objs)
    build_c_example(force=force or bool(jobs), run=run)
    return LIB


C_EXAMPLE = os.path.join(HERE, "build", "capi_conv")


def build_c_example(force=False, run=None):
    """examples/capi_conv.c: a plain-C host of the ABI (gcc, no Python / torch / C++), linked against libsed_hip.so and
    the ROCm HIP runtime.  Run by tests/test_gpu_ops.py on the GPU box."""
    src = os.path.join(os.path.dirname(HERE), "examples", "capi_conv.c")
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    if not (force or _stale(C_EXAMPLE, [src, LIB, os.path.join(INCLUDE, "sed_hip.h")])):
        return C_EXAMPLE
    cmd = [os.environ.get("CC", "gcc"), "-O2", "-std=c11", "-Wall", src, "-I", INCLUDE, "-I", os.path.join(rocm, "include"),
           "-D__HIP_PLATFORM_AMD__", "-L", HERE, "-lsed_hip", "-L", os.path.join(rocm, "lib"), "-lamdhip64", "-lm",
           "-Wl,-rpath,$ORIGIN/..", "-Wl,-rpath," + os.path.join(rocm, "lib"), "-o", C_EXAMPLE]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("C example failed to build:\n%s\n%s" % (" ".join(cmd), r.stderr))
    return C_EXAMPLE


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
