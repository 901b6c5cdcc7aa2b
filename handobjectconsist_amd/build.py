"""Build libmeshraster_hip.so in-tree with hipcc for gfx950 (no torch involved)."""
import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
SOURCES = ["raster_fwd.hip", "raster_bwd.hip", "warp.hip", "vertex_stage.hip", "mano_lbs.hip", "meshreg_post.hip",
           "frame_batch.hip", "frozen_bn.hip", "stem_pool.hip"]
LIB_PATH = os.path.join(_HERE, "libmeshraster_hip.so")
# -ffp-contract=off: parity-critical fp32 expressions must round operation by operation
# exactly like the CPU oracle (SURVEY 7 "bit-faithful coverage").
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
               "-fvisibility=hidden", "-Wall", "-Wno-unused-function"]


def _hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC or add /opt/rocm/bin to PATH)")


def needs_build():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)]
    deps.append(os.path.join(_HERE, "..", "include", "meshraster_hip.h"))
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build_library(force=False, verbose=False):
    """hipcc --offload-arch=gfx950 ... -> handobjectconsist_amd/libmeshraster_hip.so"""
    if not force and not needs_build():
        return LIB_PATH
    cmd = [_hipcc()] + HIPCC_FLAGS + ["-o", LIB_PATH + ".tmp"] + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    os.replace(LIB_PATH + ".tmp", LIB_PATH)
    return LIB_PATH


if __name__ == "__main__":
    print(build_library(force=True, verbose=True))
