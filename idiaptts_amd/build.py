"""Builds libidiaptts_amd.so (HIP kernels + C ABI) in-tree for gfx950 with hipcc.

Usage: ``python -m idiaptts_amd.build [--force]``.  hipcc cross-compiles without a GPU.
"""
import concurrent.futures
import os
import shutil
import subprocess
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_DIR = os.path.join(_HERE, "_lib")
LIB_PATH = os.path.join(LIB_DIR, "libidiaptts_amd.so")
ARCH = "gfx950"

CXXFLAGS = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=" + ARCH, "-Wall",
            "-Wno-unused-function", "-ffp-contract=on", "-munsafe-fp-atomics"]


# host-only translation units (file I/O): no device code, no contraction of a*b+c
HOST_CXXFLAGS = ["-O3", "-std=c++17", "-fPIC", "-Wall", "-ffp-contract=off", "-pthread"]


def _hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found; cannot build libidiaptts_amd.so")
    return exe


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC)
                  if f.endswith(".hip") or f.endswith(".cpp"))


def _deps():
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hdrs.append(os.path.join(_HERE, "..", "include", "idiaptts_amd.h"))
    return hdrs


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _compile(src, obj):
    flags = HOST_CXXFLAGS if src.endswith(".cpp") else CXXFLAGS
    cmd = [_hipcc()] + flags + ["-c", src, "-o", obj]
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if res.returncode != 0:
        raise RuntimeError("hipcc failed for {}:\n{}".format(src, res.stdout))
    return res.stdout


def build_library(force=False, verbose=False):
    os.makedirs(LIB_DIR, exist_ok=True)
    srcs = sources()
    hdrs = _deps()
    objs = [os.path.join(LIB_DIR, os.path.splitext(os.path.basename(s))[0] + ".o") for s in srcs]
    todo = [(s, o) for s, o in zip(srcs, objs) if force or _stale(o, [s] + hdrs)]
    if todo:
        with concurrent.futures.ThreadPoolExecutor(max_workers=min(6, len(todo))) as ex:
            for out in ex.map(lambda so: _compile(*so), todo):
                if verbose and out.strip():
                    print(out)
    if todo or _stale(LIB_PATH, objs):
        cmd = [_hipcc(), "-shared", "-fPIC", "-pthread", "--offload-arch=" + ARCH, "-o", LIB_PATH] + objs
        res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if res.returncode != 0:
            raise RuntimeError("link failed:\n" + res.stdout)
    return LIB_PATH


if __name__ == "__main__":
    path = build_library(force="--force" in sys.argv, verbose=True)
    print(path)
