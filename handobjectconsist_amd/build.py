"""Build libmeshraster_hip.so in-tree with hipcc for gfx950 (no torch involved)."""
import hashlib
import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
SOURCES = ["raster_fwd.hip", "raster_bwd.hip", "warp.hip", "vertex_stage.hip", "mano_lbs.hip", "meshreg_post.hip",
           "frame_batch.hip", "frozen_bn.hip", "stem_pool.hip", "pair_step.hip"]
LIB_PATH = os.path.join(_HERE, "libmeshraster_hip.so")
# -ffp-contract=off: parity-critical fp32 expressions must round operation by operation
# exactly like the CPU oracle (SURVEY 7 "bit-faithful coverage").
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
               "-fvisibility=hidden", "-Wall", "-Wno-unused-function"]
# profiling builds only (e.g. HOC_HIPCC_FLAGS=-DMR_WG_TIMELINE for scripts/wg_timeline.py); part of the source hash
HIPCC_FLAGS += os.environ.get("HOC_HIPCC_FLAGS", "").split()


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


OBJ_DIR = os.path.join(_HERE, "build")  # git-ignored object cache (one .o per source, keyed by content hash)
COMPILE_FLAGS = [f for f in HIPCC_FLAGS if f != "-shared"]


def _object_for(src, headers_hash):
    with open(os.path.join(CSRC, src), "rb") as fh:
        key = hashlib.sha256(" ".join(COMPILE_FLAGS).encode() + headers_hash + fh.read()).hexdigest()[:16]
    return os.path.join(OBJ_DIR, f"{os.path.splitext(src)[0]}.{key}.o")


def build_library(force=False, verbose=False):
    """hipcc --offload-arch=gfx950 ... -> handobjectconsist_amd/libmeshraster_hip.so.  Sources are compiled
    to objects in parallel (cached by content hash under build/) and linked; prints whether the library was
    compiled or an up-to-date one (same source hash) was reused."""
    if not force and not needs_build():
        if verbose:
            print("libmeshraster_hip.so: reused (source hash %s)" % source_hash()[:12])
        return LIB_PATH
    os.makedirs(OBJ_DIR, exist_ok=True)
    hh = hashlib.sha256()
    for hdr in sorted(f for f in os.listdir(CSRC) if f.endswith((".hpp", ".h"))) + [os.path.join("..", "..", "include", "meshraster_hip.h")]:
        with open(os.path.join(CSRC, hdr), "rb") as fh:
            hh.update(fh.read())
    objs = [_object_for(src, hh.digest()) for src in SOURCES]
    procs = []
    for src, obj in zip(SOURCES, objs):
        if os.path.exists(obj):
            continue
        cmd = [_hipcc()] + COMPILE_FLAGS + ["-c", "-o", obj + ".tmp.o", os.path.join(CSRC, src)]
        if verbose:
            print(" ".join(cmd))
        procs.append((obj, cmd, subprocess.Popen(cmd)))
    for obj, cmd, proc in procs:
        if proc.wait() != 0:
            raise subprocess.CalledProcessError(proc.returncode, cmd)
        os.replace(obj + ".tmp.o", obj)
    cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_PATH + ".tmp"] + objs
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    os.replace(LIB_PATH + ".tmp", LIB_PATH)
    keep = set(objs)
    for junk in os.listdir(OBJ_DIR):  # objects of older source versions
        if os.path.join(OBJ_DIR, junk) not in keep:
            os.remove(os.path.join(OBJ_DIR, junk))
    with open(STAMP_PATH, "w") as fh:
        fh.write(source_hash() + "\n")
    if verbose:
        print("libmeshraster_hip.so: compiled %d of %d sources (source hash %s)" % (len(procs), len(SOURCES), source_hash()[:12]))
    return LIB_PATH


if __name__ == "__main__":
    print(build_library(force=True, verbose=True))
