"""Build the gfx950 shared library (HIP kernels + C ABI + host mirror) in-tree with hipcc.

One object per source (compiled in parallel, rebuilt only when the source, a header or the flag string changed), then one link.
A/B builds: SIGMAZERO_LIB=<other .so path> selects the output (and the library `_native` loads); SIGMAZERO_EXTRA_FLAGS (e.g.
-DNN_ROWSKIP=0) is honoured ONLY together with SIGMAZERO_LIB, so a leftover variable can never turn the default library into an
ablated one.  The flag string a library was built with is kept next to it (<lib>.flags) and is part of the up-to-date check.
"""
import hashlib
import os
import shutil
import subprocess

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
DEFAULT_LIB = os.path.join(PKG, "libsigmazero_hip.so")
LIB = os.environ.get("SIGMAZERO_LIB") or DEFAULT_LIB      # SIGMAZERO_LIB: A/B runs of two builds of the library on one GPU box
SOURCES = ["sz_engine.hip", "sz_nn.hip", "sz_nn_split.hip", "sz_train.hip", "sz_host.cpp"]
HEADERS = [os.path.join(CSRC, "sz_chess.h"), os.path.join(CSRC, "sz_nn_common.h"), os.path.join(PKG, "..", "include", "sigmazero.h")]
# -ffp-contract=off: the UCB / prior arithmetic must round exactly like the reference's torch ops
FLAGS = ["-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function", "-Wno-int-to-pointer-cast"]


def extra_flags():
    extra = os.environ.get("SIGMAZERO_EXTRA_FLAGS", "").split()
    if extra and os.path.abspath(LIB) == os.path.abspath(DEFAULT_LIB):
        raise RuntimeError("SIGMAZERO_EXTRA_FLAGS=%r is set without SIGMAZERO_LIB: refusing to build the default library with A/B flags "
                           "(set SIGMAZERO_LIB=<other path> for a variant build, or unset SIGMAZERO_EXTRA_FLAGS)" % " ".join(extra))
    return extra


def flag_string():
    return " ".join(FLAGS + extra_flags())


def _obj_dir():
    tag = hashlib.sha1((flag_string() + "|" + os.path.abspath(LIB)).encode()).hexdigest()[:10]
    return os.path.join(PKG, "_build", tag)


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


HASH_GROUPS = {"tower": ["sz_nn.hip", "sz_nn_common.h"], "split": ["sz_nn_split.hip", "sz_nn_common.h"], "tree": ["sz_engine.hip", "sz_chess.h"]}


def source_hash(group="all"):
    """sha1 over kernel sources: ties a profile record to the code it was taken from (bench.py `traffic`).  group: "tower" (k_tower16_bf16),
    "split" (k_tower_split), "tree" (k_search_step) or "all"."""
    files = sorted([os.path.join(CSRC, s) for s in SOURCES] + HEADERS) if group == "all" else [os.path.join(CSRC, f) for f in HASH_GROUPS[group]]
    h = hashlib.sha1()
    for p in files:
        with open(p, "rb") as f:
            h.update(os.path.basename(p).encode() + b"\0" + f.read())
    return h.hexdigest()[:16]


def needs_build():
    deps = [os.path.join(CSRC, s) for s in SOURCES] + HEADERS
    if _stale(LIB, deps):
        return True
    try:
        with open(LIB + ".flags") as f:
            return f.read().strip() != flag_string()
    except OSError:
        return True


def build(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    odir = _obj_dir()
    os.makedirs(odir, exist_ok=True)
    flags = FLAGS + extra_flags()
    jobs, objs = [], []
    for s in SOURCES:
        src, obj = os.path.join(CSRC, s), os.path.join(odir, os.path.splitext(s)[0] + ".o")
        objs.append(obj)
        if force or _stale(obj, [src] + HEADERS):
            cmd = [hipcc] + flags + ["-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd))
            jobs.append((cmd, subprocess.Popen(cmd)))
    for cmd, p in jobs:
        if p.wait() != 0:
            raise subprocess.CalledProcessError(p.returncode, cmd)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", LIB]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    with open(LIB + ".flags", "w") as f:
        f.write(flag_string() + "\n")
    return LIB
