"""Build libmeshraster_hip.so in-tree with hipcc for gfx950 (no torch involved)."""
import hashlib
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


STAMP_PATH = LIB_PATH + ".srchash"  # git-ignored, travels to the GPU box next to the .so


def source_hash():
    """sha256 over the compiler flags and every file the library is built from (content, not mtimes:
    a fresh checkout or a gpurun snapshot resets mtimes)."""
    h = hashlib.sha256(" ".join(HIPCC_FLAGS).encode())
    deps = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC))
    deps.append(os.path.join(_HERE, "..", "include", "meshraster_hip.h"))
    for d in deps:
        if os.path.isfile(d):
            h.update(os.path.basename(d).encode())
            with open(d, "rb") as fh:
                h.update(fh.read())
    return h.hexdigest()


def needs_build():
    if not os.path.exists(LIB_PATH) or not os.path.exists(STAMP_PATH):
        return True
    with open(STAMP_PATH) as fh:
        return fh.read().strip() != source_hash()


def build_library(force=False, verbose=False):
    """hipcc --offload-arch=gfx950 ... -> handobjectconsist_amd/libmeshraster_hip.so.  Prints whether the
    library was compiled or an up-to-date one (same source hash) was reused."""
    if not force and not needs_build():
        if verbose:
            print("libmeshraster_hip.so: reused (source hash %s)" % source_hash()[:12])
        return LIB_PATH
    cmd = [_hipcc()] + HIPCC_FLAGS + ["-o", LIB_PATH + ".tmp"] + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    os.replace(LIB_PATH + ".tmp", LIB_PATH)
    for junk in os.listdir(_HERE):  # clang-offload-bundler leftovers of an interrupted link
        if junk.startswith("libmeshraster_hip.so.tmp") or junk.startswith("libmeshraster_hip.so.tmp."):
            os.remove(os.path.join(_HERE, junk))
    with open(STAMP_PATH, "w") as fh:
        fh.write(source_hash() + "\n")
    if verbose:
        print("libmeshraster_hip.so: compiled (source hash %s)" % source_hash()[:12])
    return LIB_PATH


if __name__ == "__main__":
    print(build_library(force=True, verbose=True))
