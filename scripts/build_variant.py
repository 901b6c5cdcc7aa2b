"""A / B libraries for kernel comparisons on ONE GPU box (box-to-box clocks differ by ~10 %): the library as it is, but with one
source file taken from another file or git revision -> handobjectconsist_amd/variants/lib_<name>.so (git-ignored *.so, travels
with the gpurun snapshot); select with HOC_LIB_PATH=handobjectconsist_amd/variants/lib_<name>.so.
    python scripts/build_variant.py <name> <source.hip> <replacement file | git-rev>"""
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from handobjectconsist_amd import build as B  # noqa: E402

name, src, repl = sys.argv[1:4]
out_dir = os.path.join(ROOT, "handobjectconsist_amd", "variants")
os.makedirs(out_dir, exist_ok=True)
tmp = tempfile.mkdtemp(prefix="hoc_variant_")
import hashlib  # noqa: E402

hh = hashlib.sha256()
for hdr in sorted(f for f in os.listdir(B.CSRC) if f.endswith((".hpp", ".h"))) + [os.path.join("..", "..", "include", "meshraster_hip.h")]:
    with open(os.path.join(B.CSRC, hdr), "rb") as fh:
        hh.update(fh.read())
objs = []
for s in B.SOURCES:
    path = os.path.join(B.CSRC, s)
    cached = B._object_for(s, hh.digest())
    if s != src and os.path.exists(cached):  # (the library's own object of an unchanged source)
        objs.append(cached)
        continue
    if s == src:
        text = open(repl).read() if os.path.isfile(repl) else subprocess.run(
            ["git", "-C", ROOT, "show", f"{repl}:handobjectconsist_amd/csrc/{s}"], capture_output=True, text=True, check=True).stdout
        path = os.path.join(tmp, s)
        open(path, "w").write(text)
    obj = os.path.join(tmp, s + ".o")
    subprocess.check_call([B._hipcc()] + B.COMPILE_FLAGS + ["-I", B.CSRC, "-c", "-o", obj, path])
    objs.append(obj)
out = os.path.join(out_dir, f"lib_{name}.so")
subprocess.check_call([B._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs)
print(out)
