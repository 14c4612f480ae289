"""Builds libvince_hip.so (all HIP kernels + the C ABI) for gfx950 with hipcc, in-tree.

    python -m vince_amd.build            # incremental
    python -m vince_amd.build --force

hipcc cross-compiles without a GPU.  The .so lands in vince_amd/lib/ (git-ignored, but shipped to the GPU box).
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
OBJDIR = os.path.join(HERE, "build")
LIB = os.path.join(LIBDIR, "libvince_hip.so")
SOURCES = ["error.cpp", "conv_igemm.hip", "conv_igemm_x3.hip", "conv_m8.hip", "conv_xjoin.hip", "conv3x3_strip.hip", "conv_wgrad.hip", "conv_wgrad_tr.hip", "conv_wgrad_x3.hip", "bn_pool.hip", "bn_gram.hip", "bn_algebra.hip", "misc.hip", "infonce.hip", "trunk.hip", "augment.hip"]
# augment.hip restates Pillow's double / float arithmetic: an fma where the C library rounds twice changes results
EXTRA_FLAGS = {"augment.hip": ["-ffp-contract=off"]}
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-Wno-unused-result"]


def _newer(a, b):
    return (not os.path.exists(b)) or os.path.getmtime(a) > os.path.getmtime(b)


def build(force=False, verbose=True, measure=False):
    """measure=True: the measurement build (-DVINCE_MEASURE: ablation switches inside the conv kernels) as
    lib/libvince_hip_measure.so, loaded by tools through VINCE_HIP_LIB; the product library never carries them."""
    global OBJDIR, LIB
    if measure:
        OBJDIR = os.path.join(HERE, "build_measure")
        LIB = os.path.join(LIBDIR, "libvince_hip_measure.so")
    extra = os.environ.get("VINCE_BUILD_FLAGS", "").split()    # A/B builds of kernel variants: a second library, by name
    if extra:
        tag = os.environ.get("VINCE_BUILD_TAG", "ab")
        OBJDIR = os.path.join(HERE, "build_" + tag)
        LIB = os.path.join(LIBDIR, "libvince_hip_%s.so" % tag)
    os.makedirs(LIBDIR, exist_ok=True)
    os.makedirs(OBJDIR, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    headers = [os.path.join(CSRC, "common.h"), os.path.join(CSRC, "conv_core.h"), os.path.join(CSRC, "conv_igemm_impl.h"), os.path.join(CSRC, "conv_wgrad.h"), os.path.join(os.path.dirname(HERE), "include", "vince_hip.h")]
    jobs = []
    objs = []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(OBJDIR, os.path.splitext(s)[0] + ".o")
        objs.append(obj)
        if force or _newer(src, obj) or any(_newer(h, obj) for h in headers):
            cmd = [hipcc] + FLAGS + (["-DVINCE_MEASURE"] if measure else []) + extra + EXTRA_FLAGS.get(s, []) + (["-x", "hip"] if s.endswith(".cpp") else []) + ["-c", src, "-o", obj]
            jobs.append(cmd)

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n%s\n%s" % (" ".join(cmd), r.stderr))
        return r

    if jobs:
        with ThreadPoolExecutor(max_workers=min(8, len(jobs))) as ex:
            list(ex.map(run, jobs))
    if jobs or not os.path.exists(LIB) or any(_newer(o, LIB) for o in objs):
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, measure="--measure" in sys.argv))
