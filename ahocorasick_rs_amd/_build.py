"""In-tree build of the native pieces (no network, no pip):

  libacx_hip.so                     hipcc --offload-arch=gfx950: HIP kernels + C ABI
  ahocorasick_rs.cpython-*.so       g++: C++ CPython extension (the host shim that
                                    mirrors the reference's PyO3 module), linked
                                    against libacx_hip.so via $ORIGIN rpath

hipcc cross-compiles gfx950 without a GPU, so this runs in the build container;
the resulting .so files travel to the GPU box with the repository snapshot.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
import sysconfig

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(ROOT, "include")
LIB = os.path.join(HERE, "libacx_hip.so")
EXT = os.path.join(HERE, "ahocorasick_rs" + (sysconfig.get_config_var("EXT_SUFFIX") or ".so"))

LIB_SOURCES = ["kernels.hip", "acx_api.cpp", "automaton.cpp", "comm.cpp"]
LIB_HEADERS = ["kernels.hpp", "automaton.hpp", "device_types.hpp", os.path.join(INCLUDE, "acx.h")]
EXT_SOURCES = ["pymodule.cpp"]


def _newer(target: str, deps) -> bool:
    if not os.path.exists(target):
        return False
    t = os.path.getmtime(target)
    return all(os.path.getmtime(d) <= t for d in deps)


def _hipcc() -> str:
    for c in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found: the HIP library cannot be built")


def build_lib(force: bool = False, verbose: bool = False) -> str:
    srcs = [os.path.join(CSRC, s) for s in LIB_SOURCES]
    deps = srcs + [h if os.path.isabs(h) else os.path.join(CSRC, h) for h in LIB_HEADERS]
    if not force and _newer(LIB, deps):
        return LIB
    cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
           "-I" + INCLUDE, "-o", LIB] + srcs + ["-ldl"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


def build_ext(force: bool = False, verbose: bool = False) -> str:
    srcs = [os.path.join(CSRC, s) for s in EXT_SOURCES]
    if not all(os.path.exists(s) for s in srcs):
        return ""
    deps = srcs + [os.path.join(INCLUDE, "acx.h"), LIB]
    if not force and _newer(EXT, deps):
        return EXT
    py_inc = sysconfig.get_paths()["include"]
    cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-fvisibility=hidden",
           "-I" + INCLUDE, "-I" + py_inc, "-o", EXT] + srcs + \
          ["-L" + HERE, "-lacx_hip", "-Wl,-rpath,$ORIGIN"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return EXT


def build_all(force: bool = False, verbose: bool = False) -> None:
    build_lib(force, verbose)
    build_ext(force, verbose)


if __name__ == "__main__":
    build_all(force="--force" in sys.argv, verbose=True)
