"""Build the gfx950 shared library in-tree (no JIT cache: the .so travels with the tree)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libzkir_hip.so")

DEVICE_SOURCES = ["hip/interp.hip"]
HOST_SOURCES = ["hip/runtime.cpp", "hip/reducer.cpp", "host/capi.cpp", "host/generator.cpp", "host/circom.cpp", "host/noir.cpp", "host/lower.cpp",
                "host/sexpr.cpp", "host/gadget.cpp", "host/fr_host.cpp"]


def _newer(src_paths, out):
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(p) > t for p in src_paths)


def build(force=False, verbose=False):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    headers = []
    for root, _, files in os.walk(CSRC):
        headers += [os.path.join(root, f) for f in files if f.endswith((".h", ".hpp", ".cuh", ".inc"))]
    headers.append(os.path.join(HERE, "..", "include", "zkir_hip.h"))
    objs = []
    for rel in DEVICE_SOURCES + HOST_SOURCES:
        src = os.path.join(CSRC, rel)
        if not os.path.exists(src):
            raise FileNotFoundError(src)
        obj = os.path.join(objdir, rel.replace("/", "_") + ".o")
        objs.append(obj)
        if force or _newer([src] + headers, obj):
            cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
                   "-c", src, "-o", obj]
            if rel.endswith(".cpp"):
                cmd.insert(1, "-x")
                cmd.insert(2, "hip")
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
    if force or _newer(objs, LIB):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-ldl"]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
