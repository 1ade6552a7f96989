"""Build libterragan_hip.so (gfx950) and, for tests only, nothing else.  Usage: python build.py [--force]"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SRC = [os.path.join(HERE, "csrc", f) for f in ("igemm.hip", "pointwise.hip", "smallconv.hip", "metrics.hip")]
# every header / include file under csrc/ is a dependency of every object (a stale .so on the GPU box is worse than a rebuild)
HDR = sorted(os.path.join(HERE, "csrc", f) for f in os.listdir(os.path.join(HERE, "csrc")) if f.endswith((".h", ".inc"))) + \
    [os.path.join(ROOT, "include", "terragan_hip.h")]
LIB = os.path.join(HERE, "lib", "libterragan_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def _stale(out, deps):
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    objs = []
    for src in SRC:
        obj = os.path.join(HERE, "lib", os.path.basename(src) + ".o")
        objs.append(obj)
        if force or _stale(obj, [src] + HDR):
            cmd = [HIPCC, "-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-I" + os.path.join(ROOT, "include"),
                   "-I" + os.path.join(HERE, "csrc"), "-Wno-comment", "-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
    if force or _stale(LIB, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
