"""Builds renderer_amd/libsxe_hip.so with hipcc for gfx950 (in-tree, so it ships with the tree)."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libsxe_hip.so")
SOURCES = ["sxe_api.hip", "sxe_geometry.hip", "sxe_raster.hip"]
HEADERS = ["sxe_internal.h", "sxe_device.h", "sxe_math.h", "sxe_srgb_tables.h", os.path.join("..", "..", "include", "sxe.h")]

# -ffp-contract=off: the kernels and the host maths must round every operation separately
# (the parity contract); gfx950 only, no other offload target.
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fPIC", "-shared", "-std=c++17",
               "-Wall", "-Wno-unused-function"]


def source_hash():
    """SHA-256 over the kernel and host sources the library is built from: counter summaries under
    profiles/ carry it, so that a bench line can tell whether they describe the kernels it ran."""
    import hashlib
    h = hashlib.sha256()
    for f in sorted(SOURCES + HEADERS):
        with open(os.path.join(CSRC, f), "rb") as fh:
            h.update(f.encode() + b"\0" + fh.read())
    return h.hexdigest()[:16]


def stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in SOURCES + HEADERS)


def build(force=False, verbose=False):
    if not force and not stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc] + HIPCC_FLAGS + ["-o", LIB] + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


HOST_DIR = os.path.join(HERE, "..", "host")
HOST_DEMO = os.path.join(HOST_DIR, "engine_demo")


def build_host(force=False, verbose=False):
    """The C++ host side (include/sxe_drawer.hpp) and its frame-loop program: plain g++, no HIP headers."""
    src = os.path.join(HOST_DIR, "engine_demo.cpp")
    deps = [src, os.path.join(HERE, "..", "include", "sxe_drawer.hpp"), os.path.join(HERE, "..", "include", "sxe.h"), LIB]
    if not force and os.path.exists(HOST_DEMO) and all(os.path.getmtime(d) <= os.path.getmtime(HOST_DEMO) for d in deps):
        return HOST_DEMO
    cmd = [os.environ.get("CXX", "g++"), "-std=c++17", "-O2", "-Wall", "-Wextra", "-I", os.path.join(HERE, "..", "include"), src,
           "-o", HOST_DEMO, "-L", HERE, "-lsxe_hip", "-Wl,-rpath,$ORIGIN/../renderer_amd", "-Wl,-rpath-link,/opt/rocm/lib"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return HOST_DEMO


if __name__ == "__main__":
    build(force=True, verbose=True)
    build_host(force=True, verbose=True)
