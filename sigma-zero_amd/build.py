"""Build the gfx950 shared library (HIP kernels + C ABI + host mirror) in-tree with hipcc."""
import os
import shutil
import subprocess

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
LIB = os.environ.get("SIGMAZERO_LIB") or os.path.join(PKG, "libsigmazero_hip.so")      # SIGMAZERO_LIB: A/B runs of two builds of the library on one GPU box
SOURCES = ["sz_engine.hip", "sz_nn.hip", "sz_host.cpp"]
HEADERS = [os.path.join(CSRC, "sz_chess.h"), os.path.join(PKG, "..", "include", "sigmazero.h")]
# -ffp-contract=off: the UCB / prior arithmetic must round exactly like the reference's torch ops
FLAGS = ["-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-fPIC", "-shared", "-std=c++17", "-Wall", "-Wno-unused-function"]


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + HEADERS
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    extra = os.environ.get("SIGMAZERO_EXTRA_FLAGS", "").split()           # A/B builds, e.g. -DNN_ROWSKIP=0 -DNN_WBUF=0
    cmd = [hipcc] + FLAGS + extra + [os.path.join(CSRC, s) for s in SOURCES] + ["-o", LIB]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB
