"""The C-ABI shared library: loads without a GPU, exports every MR_API prototype of
include/meshraster_hip.h, the ctypes table covers all of them, argument validation happens
before any device work, and the product fails loudly when the library is missing."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "meshraster_hip.h")


def header_symbols():
    src = open(HEADER).read()
    return sorted(set(re.findall(r"MR_API\s+(?:int64_t|int)\s+(mr_[a-z0-9_]+)\s*\(", src)))


def test_header_declares_the_five_upstream_entry_points():
    syms = header_symbols()
    for name in ("mr_forward_face_index_map", "mr_forward_texture_sampling", "mr_backward_pixel_map",
                 "mr_backward_textures", "mr_backward_depth_map"):
        assert name in syms
    assert len(syms) >= 15


def test_library_exports_every_declared_symbol():
    from handobjectconsist_amd import _lib

    assert os.path.exists(_lib.LIB_PATH), "run __graft_entry__.build() first"
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in header_symbols():
        assert hasattr(lib, name), f"{name} declared in the header but not exported"
    assert sorted(_lib.SIGNATURES) == header_symbols(), "ctypes table and header disagree"
    assert _lib.load().mr_abi_version() == _lib.ABI_VERSION


def test_header_prototype_arity_matches_ctypes_table():
    from handobjectconsist_amd import _lib

    src = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    for name, (_, argtypes) in _lib.SIGNATURES.items():
        m = re.search(r"MR_API\s+(?:int64_t|int)\s+" + name + r"\s*\((.*?)\)\s*;", src, flags=re.S)
        assert m, name
        params = [p for p in m.group(1).split(",") if p.strip() and p.strip() != "void"]
        assert len(params) == len(argtypes), f"{name}: header has {len(params)} parameters, ctypes {len(argtypes)}"


def test_header_prototype_types_match_ctypes_table():
    """... and position by position the KIND of every parameter: pointer / int / int64 / float (a swapped int and float
    would pass the arity check and reinterpret bits at run time), plus the return type."""
    from handobjectconsist_amd import _lib

    def kind(decl):
        decl = decl.strip()
        if "*" in decl or re.search(r"\bmr_stream_t\b", decl):
            return ctypes.c_void_p
        words = decl.split()
        base = [w for w in words[:-1] if w not in ("const", "unsigned")] or words  # (last word = the parameter's name)
        t = base[0]
        return {"int": ctypes.c_int, "int64_t": ctypes.c_int64, "float": ctypes.c_float}[t]

    src = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    for name, (res, argtypes) in _lib.SIGNATURES.items():
        m = re.search(r"MR_API\s+(int64_t|int)\s+" + name + r"\s*\((.*?)\)\s*;", src, flags=re.S)
        assert m, name
        assert res is {"int": ctypes.c_int, "int64_t": ctypes.c_int64}[m.group(1)], f"{name}: return type"
        params = [p for p in m.group(2).split(",") if p.strip() and p.strip() != "void"]
        for i, (decl, want) in enumerate(zip(params, argtypes)):
            assert kind(decl) is want, f"{name}: parameter {i} `{' '.join(decl.split())}` is bound as {want.__name__}"


def test_pair_step_struct_layout_matches_the_binding():
    """ABI 8: the one argument block of mr_pair_step_* as the library lays it out (size, offset of every field in declaration
    order) against the ctypes Structure of warping/pairstep.py and against the header's field list; sizes need no device."""
    from handobjectconsist_amd import _lib
    from handobjectconsist_amd.warping import pairstep

    lib = _lib.load()
    assert ctypes.sizeof(pairstep.MrPairStep) == lib.mr_pair_step_struct_bytes()
    fields = [f for f, _ in pairstep.MrPairStep._fields_]
    offs = (ctypes.c_int64 * 128)()
    assert lib.mr_pair_step_field_offsets(offs, 128) == len(fields)
    assert [getattr(pairstep.MrPairStep, f).offset for f in fields] == list(offs[:len(fields)])
    src = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    body = re.search(r"typedef struct MrPairStep \{(.*?)\} MrPairStep;", src, flags=re.S).group(1)
    decl_names = re.findall(r"([A-Za-z_][A-Za-z0-9_]*)\s*(?:,|;)", body)
    assert decl_names == fields, "header and binding list the fields in different orders"
    st = pairstep.MrPairStep()
    for k, v in dict(batch_size=64, num_verts_a=778, num_verts_b=1002, num_hand_faces=1552, num_obj_faces=2000, fill_back=1,
                     image_size=256, height=256, width=256, jitter_channels=3).items():
        setattr(st, k, v)
    sc, sv, th = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int64()
    assert lib.mr_pair_step_sizes(ctypes.byref(st), ctypes.byref(sc), ctypes.byref(sv), ctypes.byref(th)) == 0
    px = 128 * 256 * 256
    assert sv.value >= px * (4 + 12 + 12 + 8) and sc.value >= px * 4 * 6 and th.value == px * 4
    # (every raster the fused path takes: sizes only, whatever the planes' sizes are modulo the regions' 256-byte alignment --
    # a region aliased onto two others refused 36 x 36 and friends for an hour of round 6, found by tests/test_gpu_fuzz.py)
    for B, is_, h, w in ((1, 12, 12, 12), (3, 36, 36, 20), (2, 100, 64, 100), (5, 132, 132, 132), (8, 480, 270, 480), (1, 4, 4, 4)):
        st.batch_size, st.image_size, st.height, st.width = B, is_, h, w
        assert lib.mr_pair_step_sizes(ctypes.byref(st), ctypes.byref(sc), ctypes.byref(sv), ctypes.byref(th)) == 0, (B, is_)
        assert sc.value >= 2 * B * is_ * is_ * (4 * 6 + 16) and 0 <= th.value - 2 * B * is_ * is_ * 4 < 256 and th.value % 256 == 0
    st.batch_size = 64
    st.image_size = 258  # (not a multiple of 4: the fused path does not apply)
    st.height = st.width = 258
    assert lib.mr_pair_step_sizes(ctypes.byref(st), ctypes.byref(sc), ctypes.byref(sv), ctypes.byref(th)) == -2
    assert lib.mr_pair_step_forward(None, None) == -1 and lib.mr_pair_step_backward(None, None) == -1


def test_argument_validation_needs_no_device():
    from handobjectconsist_amd import _lib

    lib = _lib.load()
    assert lib.mr_render_workspace_bytes(-1, 10, 64) == -1
    assert lib.mr_render_workspace_bytes(2, 100, 64) >= 2 * 100 * (16 + 48)
    assert lib.mr_pair_consist_workspace_bytes(4, 256, 256) == 4 * 4 * 64 * 16
    assert lib.mr_pair_consist_workspace_bytes(1, 0, 5) == -1
    # raster backward scratch (round 3): flags + counts + owner list + per-image owner records, 37 bytes per face and
    # 20 per image up to the 256-byte alignment of its parts; independent of the raster size (no packed map copies)
    w = lib.mr_render_backward_list_workspace_bytes(64, 3076)
    assert 37 * 64 * 3076 <= w <= 37 * 64 * 3076 + 20 * 64 + 5 * 256
    # ... + kernel D's strip bookkeeping (round 4): a weight per (image, axis, strip of 4 lines at 256 x 256) and the
    # per-XCD lists of the strips with work, 4 bytes each
    full = lib.mr_render_backward_workspace_bytes(64, 3076, 256)
    assert w + 2 * 64 * 128 * 4 <= full <= w + 2 * 64 * 128 * 4 + 3 * 256
    assert lib.mr_render_backward_workspace_bytes(-1, 1, 8) == -1
    null = ctypes.c_void_p(None)
    # NULL pointers / bad sizes are rejected with MR_ERR_BADARG before anything touches HIP
    assert lib.mr_warp_forward(null, null, null, null, 1, 3, 8, 8, 0.99999, 0, null) == -1
    assert lib.mr_backward_textures(null, null, null, null, null, 1, 1, 8, 2, null) == -1
    assert lib.mr_face_inv_map(null, null, null, 1, 1, 8, null) == -1
    with pytest.raises(RuntimeError, match="bad argument"):
        _lib.call("mr_occlusion_mask", null, null, null, null, 0, null, null, null, null, 1, 8, 8, 0.03, 0.99999, null)


def test_missing_library_fails_loudly(monkeypatch):
    from handobjectconsist_amd import _lib

    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", os.path.join(ROOT, "does_not_exist.so"))
    with pytest.raises(RuntimeError, match="no CPU / PyTorch fallback"):
        _lib.load()


def test_cpu_tensors_are_rejected_like_the_reference():
    import torch

    from handobjectconsist_amd.neurender import rasterize
    from handobjectconsist_amd.warping import imgflowarp

    with pytest.raises(TypeError):
        rasterize.rasterize_rgbad(torch.zeros(1, 2, 3, 3), torch.zeros(1, 2, 2, 2, 2, 3), 8, False)
    with pytest.raises(TypeError):
        rasterize.Rasterize(8, 0.1, 100, 1e-3, (0, 0, 0), True, True, True)(torch.zeros(1, 2, 3, 3),
                                                                            torch.zeros(1, 2, 2, 2, 2, 3))
    with pytest.raises(TypeError):
        imgflowarp.warp(torch.zeros(1, 3, 4, 4), torch.zeros(1, 2, 4, 4))


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "handobjectconsist_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp")):
                txt = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", txt, flags=re.M), f
                assert "liboracle" not in txt, f


def test_call_switches_to_the_device_of_its_stream(monkeypatch):
    """``_lib.call`` launches on the device its stream argument was taken from, not on whatever device
    happens to be current (tensors on cuda:1 while cuda:0 is current)."""
    import contextlib

    import torch

    from handobjectconsist_amd import _lib

    events = []

    class FakeLib:
        @staticmethod
        def mr_fake(*args):
            events.append(("launch", torch.cuda.current_device()))
            return 0

    state = {"current": 0}

    @contextlib.contextmanager
    def fake_device(idx):
        prev, state["current"] = state["current"], idx
        events.append(("enter", idx))
        try:
            yield
        finally:
            state["current"] = prev
            events.append(("exit", idx))

    monkeypatch.setattr(_lib, "load", lambda: FakeLib)
    monkeypatch.setattr(torch.cuda, "current_device", lambda: state["current"])
    monkeypatch.setattr(torch.cuda, "device", fake_device)
    s1 = _lib._StreamArg(0)
    s1.device_index = 1
    _lib.call("mr_fake", None, 3, s1)
    assert events == [("enter", 1), ("launch", 1), ("exit", 1)]
    events.clear()
    s0 = _lib._StreamArg(0)
    s0.device_index = 0
    _lib.call("mr_fake", None, 3, s0)
    assert events == [("launch", 0)]
