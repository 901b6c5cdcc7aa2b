"""Host-side logic (no GPU): the PyTorch pieces around the kernels against the oracle's numpy
restatements / golden vectors, the lambda schedule closed forms, the lazy render dict."""
import os

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

from handobjectconsist_amd.utils import synth
from oracle import raster_ref as R

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_lambda_schedule_closed_forms():
    """warpreg.py:103-110 with trainmeshwarp.py defaults."""
    from handobjectconsist_amd.models.warpreg import consist_lambdas

    assert consist_lambdas(0, 0.999, 0.001) == (0.999, 0.0)
    ld, lc = consist_lambdas(500, 0.999, 0.001)
    assert abs(lc - 0.0005) < 1e-12 and abs(ld - 0.9985) < 1e-12
    for step in (1000, 5000):
        ld, lc = consist_lambdas(step, 0.999, 0.001)
        assert abs(lc - 0.001) < 1e-15 and abs(ld - 0.998) < 1e-12
    assert consist_lambdas(3, 1, 1, progressive_consist=False) == (1, 1)


def test_closed_faces_constants():
    from handobjectconsist_amd.models import manoutils

    closed, ignore = manoutils.get_closed_faces(torch.zeros(1538, 3, dtype=torch.long))
    assert closed.shape == (1552, 3) and ignore == list(range(1538, 1552))
    assert closed[1538].tolist() == [92, 38, 122] and closed[-1].tolist() == [214, 215, 121]
    # the reference's zero-argument call (manoutils.py:6, warpreg.py:61)
    closed0, ignore0 = manoutils.get_closed_faces()
    assert closed0.shape == (1552, 3) and ignore0 == ignore and torch.equal(closed0[1538:], closed[1538:])


def test_sample_keys_enum_and_string():
    """warpbranch reads samples keyed by the reference's Queries enums or by this package's strings."""
    import enum

    from handobjectconsist_amd.datasets import queries

    assert queries.TransQueries.JITTERMASK.value == 14 and queries.BaseQueries.HANDVERTS3D.value == 8  # auto() order
    foreign = enum.Enum("TransQueries", ["CAMINTR", "IMAGE"])  # e.g. meshreg.datasets.queries.TransQueries
    for sample in ({"image": 1}, {queries.TransQueries.IMAGE: 1}, {foreign.IMAGE: 1}):
        assert queries.lookup(sample, "image") == 1
    # GT vertices are BaseQueries entries; the augmented TransQueries entry of the same name is another tensor
    s = {queries.TransQueries.HANDVERTS3D: "trans", queries.BaseQueries.HANDVERTS3D: "base"}
    assert queries.lookup(s, "handverts3d") == "base"
    with pytest.raises(KeyError):
        queries.lookup({}, "camintr")


def test_projection_and_gather_match_oracle():
    from handobjectconsist_amd.neurender import nr_ops

    s = synth.random_scene(3, seed=1, image_size=128)
    dist = np.array([[0.1, -0.05, 0.001, 0.002, 0.01]], np.float32)
    Rm = np.array([[[0.96, -0.28, 0], [0.28, 0.96, 0], [0, 0, 1]]], np.float32)
    tv = np.array([[0.01, -0.02, 0.03]], np.float32)
    ref = R.nr_projection(s["verts1"], s["K1"], Rm, tv, dist, 128)
    got = nr_ops.projection(torch.from_numpy(s["verts1"]), torch.from_numpy(s["K1"]), torch.from_numpy(Rm),
                            torch.from_numpy(tv), torch.from_numpy(dist), 128).numpy()
    assert np.abs(got - ref).max() < 1e-5
    f = nr_ops.vertices_to_faces(torch.from_numpy(ref), torch.from_numpy(s["faces"])).numpy()
    assert np.array_equal(f, R.nr_vertices_to_faces(ref, s["faces"]))


def test_vertex_textures_proj2d_catmesh_match_oracle():
    from handobjectconsist_amd.utils import catmesh, project, textutils

    rng = np.random.default_rng(0)
    faces = rng.integers(0, 9, (2, 7, 3))
    cols = rng.standard_normal((2, 9, 3)).astype(np.float32)
    c_t = torch.from_numpy(cols).requires_grad_(True)
    tex = textutils.batch_vertex_textures(torch.from_numpy(faces), c_t)
    assert np.array_equal(tex.detach().numpy(), R.batch_vertex_textures(faces, cols))
    tex.sum().backward()
    assert c_t.grad.shape == cols.shape  # differentiable w.r.t. the vertex colours
    v = rng.standard_normal((2, 9, 3)).astype(np.float32) + [0, 0, 3]
    K = np.tile(np.array([[300.0, 0, 120], [0, 305.0, 130], [0, 0, 1]], np.float32), (2, 1, 1))
    got = project.batch_proj2d(torch.from_numpy(v.astype(np.float32)), torch.from_numpy(K)).numpy()
    assert np.abs(got - R.batch_proj2d(v, K)).max() < 1e-4
    vs, fs, _ = catmesh.batch_cat_meshes([torch.zeros(2, 4, 3), torch.ones(2, 5, 3)],
                                         [torch.zeros(2, 3, 3, dtype=torch.long), torch.ones(2, 2, 3, dtype=torch.long)])
    assert vs.shape == (2, 9, 3) and fs[:, 3:].min() == 5 and fs[:, :3].max() == 0


def test_fill_back_and_lighting_match_oracle():
    from handobjectconsist_amd.neurender import nr_ops
    from handobjectconsist_amd.neurender.renderer import Renderer

    rng = np.random.default_rng(2)
    fidx = rng.integers(0, 6, (2, 5, 3))
    tex = rng.standard_normal((2, 5, 2, 2, 2, 3)).astype(np.float32)
    f2, t2 = R.fill_back(fidx, tex)
    assert np.array_equal(Renderer._fill_back_faces(torch.from_numpy(fidx)).numpy(), f2)
    assert np.array_equal(Renderer._fill_back_textures(torch.from_numpy(tex)).numpy(), t2)
    faces = rng.standard_normal((2, 5, 3, 3)).astype(np.float32)
    lit = nr_ops.lighting(torch.from_numpy(faces), torch.from_numpy(tex), 0.8, 0.5, (1, 1, 1), (1, 0.9, 0.8), (0, 1, 0))
    assert np.abs(lit.numpy() - R.nr_lighting(faces, tex, 0.8, 0.5, (1, 1, 1), (1, 0.9, 0.8), (0, 1, 0))).max() < 1e-6


def test_look_at_and_perspective():
    from handobjectconsist_amd.neurender import nr_ops

    v = torch.tensor([[[0.0, 0.0, 0.0], [0.5, 0.25, 0.1]]])
    eye = [0, 0, -(1.0 / np.tan(np.radians(30)) + 1)]
    out = nr_ops.look_at(v, eye)
    # looking from -z towards the origin: x, y unchanged, z shifted by the eye distance
    assert torch.allclose(out[0, 0], torch.tensor([0.0, 0.0, float(-eye[2])], dtype=torch.float32), atol=1e-6)
    assert torch.allclose(out[0, 1, :2], v[0, 1, :2], atol=1e-6)
    out2 = nr_ops.look(v, eye, [0, 0, 1])
    assert torch.allclose(out, out2, atol=1e-6)
    p = nr_ops.perspective(out, angle=30)
    w = np.tan(np.radians(30))
    assert abs(float(p[0, 1, 0]) - float(out[0, 1, 0] / out[0, 1, 2]) / w) < 1e-6


def test_golden_recover_3d_proj_and_masked_mean():
    from handobjectconsist_amd.models.synthnet import recover_3d_proj
    from handobjectconsist_amd.optim import lossutils, pyramidloss

    g = np.load(os.path.join(GOLDEN, "warp_misc.npz"))
    t = torch.from_numpy
    r3d, c3d = recover_3d_proj(t(g["objpoints3d"]), t(g["camintr"]), t(g["est_scale"]), t(g["est_trans"]), off_z=0.4,
                               input_res=(256, 256))
    assert np.abs(r3d.numpy() - g["recons3d"]).max() < 1e-6 and np.abs(c3d.numpy() - g["est_c3d"]).max() < 1e-6
    mm = lossutils.batch_masked_mean_loss(t(g["dists"]), t(g["mask"]))
    assert np.abs(mm.numpy() - g["masked_mean"]).max() < 1e-6
    crit = pyramidloss.PyramidCriterion("l1")
    _, _, losses, diffs, _ = crit.compute(t(g["dists"]), torch.zeros_like(t(g["dists"])), mask=t(g["mask"]))
    assert np.abs(losses.numpy() - g["masked_mean"]).max() < 1e-6 and diffs[0].shape == g["dists"].shape
    with pytest.raises(ValueError):
        pyramidloss.PyramidCriterion("huber")
    with pytest.raises(NotImplementedError):
        pyramidloss.PyramidCriterion("ssim")


def test_renderer_argument_errors():
    from handobjectconsist_amd.neurender.renderer import Renderer

    with pytest.raises(ValueError):
        Renderer(camera_mode="orthographic")
    ren = Renderer(image_size=8, K=np.eye(3, dtype=np.float32)[None], R=np.eye(3, dtype=np.float32)[None],
                   t=np.zeros((1, 3), np.float32))
    assert ren.rasterizer_eps == 1e-3 and ren.dist_coeffs.shape == (1, 5)
    with pytest.raises(ValueError):
        ren(torch.zeros(1, 3, 3), torch.zeros(1, 1, 3, dtype=torch.long), mode="normals")


def test_lazy_render_dict():
    from handobjectconsist_amd.neurender.rasterize import _RenderOutput

    calls = []
    d = _RenderOutput({"rgb": 1, "face_inv_map": None})
    d._thunk = lambda: calls.append(1) or "materialised"
    assert set(d.keys()) == {"rgb", "face_inv_map"} and calls == []
    assert d["rgb"] == 1 and calls == []
    assert d["face_inv_map"] == "materialised" and d.get("face_inv_map") == "materialised" and calls == [1]


def test_synthetic_meshes_have_the_reference_sizes():
    hv, hf = synth.hand_template()
    ov, of = synth.object_template()
    assert hv.shape == (778, 3) and hf.shape == (1552, 3) and ov.shape == (1002, 3) and of.shape == (2000, 3)
    s = synth.random_scene(2, seed=0)
    assert s["verts1"].shape == (2, 1780, 3) and s["faces"].shape == (2, 3552, 3) and s["faces"].max() == 1779
    # the object is closed and consistently oriented: every directed edge appears exactly once
    e = np.concatenate([of[:, [0, 1]], of[:, [1, 2]], of[:, [2, 0]]])
    assert len({tuple(x) for x in e}) == len(e) and {tuple(x[::-1]) for x in e} == {tuple(x) for x in e}
    # the hand's last 14 faces are manoutils' wrist-closing fan, a few centimetres across
    from handobjectconsist_amd.models import manoutils
    assert hf[1538:].tolist() == manoutils.CLOSE_FACES
    assert np.ptp(hv[np.unique(hf[1538:])], axis=0).max() < 0.06


def test_synthetic_network_matches_the_reference_parameter_count():
    from handobjectconsist_amd.models.synthnet import SynthMeshRegNet

    m = SynthMeshRegNet()
    assert sum(p.numel() for p in m.parameters() if p.requires_grad) == 11981157
    from handobjectconsist_amd.netscripts.epochpassconsist import SyntheticConsistLoader

    ld = SyntheticConsistLoader(2, 64, device="cpu", pool=1)
    data, consist = ld.step_batches(0)
    assert data["supervision"] == "data" and consist["supervision"] == "consist" and len(consist["data"]) == 2
    loss, res, losses = m(data["data"][0])
    assert loss.shape == (1,) and res["recov_handverts3d"].shape == (2, 778, 3)
    assert res["recov_objverts3d"].shape == (2, 1002, 3) and "mano_reg_loss" in losses
    loss.backward()
    assert all(p.grad is not None for p in m.parameters() if p.requires_grad)


def test_extend_collate_pads_by_cyclic_repetition():
    """collate.py:15-36,77-83: per-mesh arrays padded to the batch maximum by repeating rows."""
    from handobjectconsist_amd.utils import collate

    a = {"objverts3d": np.arange(12, dtype=np.float32).reshape(4, 3), "objfaces": np.array([[0, 1, 2], [0, 2, 3]]),
         "image": torch.zeros(3, 4, 4), "name": "a"}
    b = {"objverts3d": np.arange(21, dtype=np.float32).reshape(7, 3) + 100, "objfaces": np.array([[0, 1, 2], [2, 3, 4], [4, 5, 6]]),
         "image": torch.ones(3, 4, 4), "name": "b"}
    out = collate.extend_collate([dict(a), dict(b)], ["objverts3d", "objfaces", "missing"])
    assert out["objverts3d"].shape == (2, 7, 3) and out["objfaces"].shape == (2, 3, 3) and out["image"].shape == (2, 3, 4, 4)
    assert torch.equal(out["objverts3d"][0, 4:], out["objverts3d"][0, :3])      # rows 0..2 again
    assert torch.equal(out["objfaces"][0, 2], out["objfaces"][0, 0])
    assert torch.equal(out["objverts3d"][1], torch.from_numpy(b["objverts3d"]))
    assert out["name"] == ["a", "b"]
    seq = collate.seq_extend_collate([[dict(a), dict(b)], [dict(b), dict(a)]], ["objverts3d", "objfaces"])
    assert len(seq) == 2 and seq[0]["objverts3d"].shape == (2, 7, 3) and seq[1]["objfaces"].shape == (2, 3, 3)
    with pytest.raises(ValueError):
        collate.seq_extend_collate([[dict(a)], [dict(a), dict(b)]])


@pytest.mark.parametrize("fused", [True, False])
def test_train_step_nan_guard_keeps_parameters_and_optimizer_state_clean(fused):
    """A NaN loss must not reach the parameters or the Adam state and must raise (reference
    epochpassconsist.py:61-63).  With a fused optimiser the guard is the optimiser's device-side `found_inf`
    operand and the ValueError comes when the next step starts; otherwise the check is synchronous."""
    from handobjectconsist_amd.netscripts import epochpassconsist as E

    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.w = torch.nn.Parameter(torch.ones(3))
            self.bad = False

        def forward(self, batch):
            loss = (self.w * batch["x"]).sum().reshape(1)
            if self.bad:
                loss = loss * float("nan")
            return loss, {"l": loss.detach()}, None, None

    net = Net()
    opt = torch.optim.Adam(net.parameters(), lr=0.1, fused=fused)
    batch = {"data": [{}], "x": torch.ones(3)}
    E.train_step([batch], net, opt)
    w1 = net.w.detach().clone()
    assert not torch.equal(w1, torch.ones(3))
    net.bad = True
    with pytest.raises(ValueError, match="nan"):
        E.train_step([batch], net, opt)   # synchronous check: raises here
        net.bad = False
        E.train_step([batch], net, opt)   # device-side guard: raises when the next step starts
    assert torch.equal(net.w.detach(), w1), "the diverged step touched the parameters"
    assert all(float(st["step"]) == 1.0 for st in opt.state.values())
    # epoch_pass flushes a pending flag at the end of the epoch
    net2, net2_opt = Net(), None
    net2_opt = torch.optim.Adam(net2.parameters(), lr=0.1, fused=fused)
    net2.bad = True
    with pytest.raises(ValueError, match="nan"):
        E.epoch_pass([batch], net2, net2_opt, loader_nb=1)
    assert torch.equal(net2.w.detach(), torch.ones(3))


def test_cat_or_view_returns_a_view_only_for_consecutive_slices():
    from handobjectconsist_amd.models.synthnet import cat_or_view

    buf = torch.arange(48.0).reshape(6, 2, 4)
    a, b, c = buf[:2], buf[2:4], buf[4:]
    v = cat_or_view([a, b, c])
    assert v.data_ptr() == buf.data_ptr() and torch.equal(v, buf)
    v = cat_or_view([b, c])
    assert v.data_ptr() == b.data_ptr() and torch.equal(v, buf[2:])
    for parts in ([a, c], [b, a], [a, b.clone()], [a, b.double()], [a, buf[2:4, :1]]):
        if parts[1].dtype == a.dtype and parts[1].shape[1:] == a.shape[1:]:
            assert torch.equal(cat_or_view(parts), torch.cat(parts))
            assert cat_or_view(parts).data_ptr() != a.data_ptr() or parts[0] is not a
    # the synthetic loader lays the frames of a step out that way
    from handobjectconsist_amd.netscripts.epochpassconsist import SyntheticConsistLoader

    loader = SyntheticConsistLoader(2, 32, seed=0, device="cpu", pool=1)
    data, consist = loader.step_batches(0)
    frames = [data["data"][0]["image"], consist["data"][0]["image"], consist["data"][1]["image"]]
    v = cat_or_view(frames)
    assert v.shape[0] == 6 and v.data_ptr() == frames[0].data_ptr() and torch.equal(v, torch.cat(frames))
    assert torch.equal(frames[0], frames[2])  # the data frame and the annotated reference are the same image


class _AlphaSpy(torch.autograd.Function):
    """alpha as given; any gradient that reaches it is handed to the vertices in full (sum over the image), so a leak
    through alpha cannot hide behind a flat region of a smooth stand-in."""

    @staticmethod
    def forward(ctx, alpha, verts):
        ctx.shape = verts.shape
        return alpha.clone()

    @staticmethod
    def backward(ctx, g):
        return None, g.sum((1, 2))[:, None, None].expand(ctx.shape).clone()


class _SoftRenderer:
    """Differentiable stand-in for ``Renderer.__call__`` on CPU (the real one needs the HIP library): a disc whose
    colour depends smoothly on the vertices and whose alpha passes every incoming gradient on to them."""
    no_light, camera_mode = True, "projection"

    def __init__(self, size):
        self.size = size
        ys, xs = torch.meshgrid(torch.arange(size, dtype=torch.float32), torch.arange(size, dtype=torch.float32), indexing="ij")
        self.grid = torch.stack([xs, ys], -1)

    def __call__(self, verts, faces, textures, K=None, detach_renders=False):
        v = verts.detach() if detach_renders else verts
        c = v.detach()[:, :, :2].mean(1) * 2 + self.size / 2                       # [B,2] disc centre
        hard = (((self.grid[None] - c[:, None, None]) ** 2).sum(-1) < (self.size / 3.0) ** 2).float()
        alpha = _AlphaSpy.apply(hard, v) if v.requires_grad else hard
        colour = textures.mean((1, 2, 3, 4))                                       # [B,3]
        rgb = colour[:, :, None, None] * hard[:, None] * (1 + 0.01 * v[:, :, 2].mean(1))[:, None, None, None]
        fim = torch.where(hard > 0.5, 0, -1).to(torch.int32)
        return {"rgb": rgb, "alpha": alpha, "face_index_map": fim}


def test_general_flow_path_keeps_the_second_alpha_out_of_autograd(monkeypatch):
    """opticalflow.py:137-139 assigns ``mask_flow2 = renderout["alpha"]`` INSIDE ``torch.no_grad()``: with attached
    renders the product ``flow21 * (alpha2 * occl2)`` must not backpropagate through alpha2.  Compared against the
    reference's statement order written out here (the occlusion check, a no-grad kernel call in the product, is the
    numpy oracle's on this CPU run)."""
    from handobjectconsist_amd.utils import project, textutils
    from handobjectconsist_amd.warping import imgflowarp, opticalflow
    from oracle import warp_ref

    def occlusion_on_cpu(m1, m2, f12, f21):
        o1, o2 = warp_ref.get_occlusion_mask(*[t_.detach().numpy() for t_ in (m1, m2, f12, f21)])
        return torch.from_numpy(np.ascontiguousarray(o1)), torch.from_numpy(np.ascontiguousarray(o2))

    monkeypatch.setattr(imgflowarp, "get_occlusion_mask", occlusion_on_cpu)

    torch.manual_seed(3)
    B, V, size = 2, 12, 16
    faces = torch.randint(0, V, (B, 7, 3))
    K = torch.tensor([[[20.0, 0, 8], [0, 20.0, 8], [0, 0, 1]]]).repeat(B, 1, 1)
    base = torch.randn(B, V, 3) * 0.3 + torch.tensor([0.0, 0.0, 2.0])
    rend = _SoftRenderer(size)

    def reference_order(v1, v2):
        p1, p2 = project.batch_proj2d(v1, K), project.batch_proj2d(v2, K)
        d12 = p2 - p1
        ro = rend(v1, faces, textutils.batch_vertex_textures(faces, torch.cat([d12, torch.ones_like(d12[:, :, :1])], -1)),
                  K=K, detach_renders=False)
        m1 = (ro["alpha"].unsqueeze(1) > 0.99999).float()
        f12 = ro["rgb"] * m1
        d21 = p1 - p2
        ro = rend(v2, faces, textutils.batch_vertex_textures(faces, torch.cat([d21, torch.ones_like(d21[:, :, :1])], -1)),
                  K=K, detach_renders=False)
        m2 = (ro["alpha"].unsqueeze(1) > 0.99999).float()
        f21 = ro["rgb"] * m2
        with torch.no_grad():
            m2 = ro["alpha"].unsqueeze(1)
            o1, o2 = imgflowarp.get_occlusion_mask(m1, m2, f12, f21)
        f12, f21 = f12 * (m1 * o1.unsqueeze(1)), f21 * (m2 * o2.unsqueeze(1))
        return [f.permute(0, 2, 3, 1)[:, :, :, :2] for f in (f12, f21)]

    grads = []
    for fn in (reference_order,
               lambda a, b: opticalflow.get_opticalflow([a, b], faces, [K, K], rend, mask_occlusions=True,
                                                        detach_textures=False, detach_renders=False)):
        v1, v2 = base.clone().requires_grad_(True), (base + 0.05).clone().requires_grad_(True)
        f12, f21 = fn(v1, v2)
        assert f21.abs().sum() > 0
        ((f12 * 0.7).sum() + (f21 * 1.3).sum()).backward()
        grads.append((v1.grad.clone(), v2.grad.clone(), f12.detach(), f21.detach()))
    for a, b in zip(*grads):
        assert torch.allclose(a, b, rtol=1e-6, atol=1e-7), float((a - b).abs().max())


def test_bench_plain_script_with_gpus_n_becomes_a_launcher(monkeypatch):
    """`python bench.py --gpus N` without RANK / WORLD_SIZE re-executes itself under torch.distributed.run with N ranks on
    127.0.0.1 and the original arguments; with fewer devices than N it exits with a one-line message (no traceback)."""
    import sys

    import bench

    launched = {}
    monkeypatch.setattr(bench.os, "execv", lambda exe, cmd: launched.update(exe=exe, cmd=list(cmd)))
    monkeypatch.setattr(bench.torch.cuda, "device_count", lambda: 8)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8", "--steps", "3", "--warmup", "1"])
    monkeypatch.delenv("HOC_SHARE_GPU", raising=False)
    args = bench.parse()
    bench.self_launch(args)
    cmd = launched["cmd"]
    assert launched["exe"] == sys.executable and cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and "--nproc-per-node=8" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and 1024 < int(cmd[cmd.index("--master-port") + 1]) < 65536
    script = cmd.index(os.path.abspath(bench.__file__))
    assert cmd[script + 1:] == ["--gpus", "8", "--steps", "3", "--warmup", "1"]
    # too few devices: a message, not an assertion
    monkeypatch.setattr(bench.torch.cuda, "device_count", lambda: 2)
    with pytest.raises(SystemExit) as e:
        bench.self_launch(args)
    assert "--gpus 8 asks for 8 devices, this node shows 2" in str(e.value)
    monkeypatch.setattr(bench.torch.cuda, "device_count", lambda: 0)
    with pytest.raises(SystemExit) as e:
        bench.self_launch(args)
    assert "needs a GPU" in str(e.value)


def test_bench_contract_line_is_compact_whatever_the_run_produced():
    """What bench.py prints on stdout is the contract line alone: <= 4096 bytes (round 5's 21 KB line was not parsed by the
    driver), carrying roofline.frac and cpu_baseline.value, strings <= 200 characters -- on round 5's full record (the
    committed details of that run) and on the same record inflated with prose, 8 ranks and 50 more kernel groups."""
    import copy
    import json

    import bench

    with open(os.path.join(ROOT, "profiles", "r05_bench_line.json")) as fh:
        full = json.load(fh)
    assert len(json.dumps(full)) > 20000
    fat = copy.deepcopy(full)
    fat["unit"] = fat["unit"] + " prose" * 400
    fat["step_mode"] = "x" * 3000
    fat["config"]["workload"] = fat["config"]["workload"] * 10
    fat["cpu_baseline"]["sample"] = fat["cpu_baseline"]["sample"] * 10
    fat["roofline"]["kernel"] = fat["roofline"]["kernel"] * 20
    fat["roofline"]["device_kernels"] = ["k" * 60] * 30
    fat["roofline_forward"]["device_kernels"] = ["k" * 60] * 30
    fat["ranks"] = {"backend": "rccl", "world_size": 8, "reducer": "r" * 500, "grad_allreduce_MB": 47.9, "bucket_MB": 16,
                    "per_rank": [{"rank": r, "device": r, "ms_per_step": 26.0 + r} for r in reversed(range(8))]}
    for i in range(50):
        fat["kernels"]["extra group %d" % i] = dict(next(iter(full["kernels"].values())))
    for rec, name in ((full, "round 5"), (fat, "inflated")):
        line = bench.contract_line(rec)
        text = json.dumps(line)
        assert len(text.encode()) <= bench.CONTRACT_LINE_MAX == 4096, (name, len(text))
        back = json.loads(text)
        for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                  "vs_baseline", "dtype", "data", "config"):
            assert k in back, (name, k)
        assert back["value"] == rec["value"] and back["ms_per_step"] == rec["ms_per_step"]
        assert back["roofline"]["frac"] == rec["roofline"]["frac"] and back["roofline"]["bound"] == "hbm"
        assert back["roofline"]["forward"]["frac"] == rec["roofline_forward"]["frac"]
        assert back["roofline"]["traffic"] == rec["roofline"]["traffic"]
        assert back["cpu_baseline"]["value"] == rec["cpu_baseline"]["value"] and back["cpu_baseline"]["cores"] == rec["cpu_baseline"]["cores"]
        assert len(back["unit"]) <= 200 and len(back["cpu_baseline"]["sample"]) <= 200
        assert "kernels" not in back and "warp_tiles" not in back and "in_step_kernels_us" not in back

        def strings(o):
            if isinstance(o, dict):
                for v in o.values():
                    yield from strings(v)
            elif isinstance(o, list):
                for v in o:
                    yield from strings(v)
            elif isinstance(o, str):
                yield o
        assert max(len(t) for t in strings(back)) <= 260, name
    assert bench.contract_line(fat)["ranks"]["ms_per_step_by_rank"] == [26.0 + r for r in range(8)]
    # legs that did not run (N > 1, --no-kernel-bench): nulls, not KeyErrors
    bare = dict(full, roofline=None, roofline_forward=None, kernels=None, cpu_baseline=None, warp_tiles=None)
    line = bench.contract_line(bare)
    assert line["roofline"] is None and line["cpu_baseline"] is None and line["value"] == full["value"]


def test_cpu_baseline_workers_are_processes_behind_a_common_start():
    """bench.py's cpu_baseline: one worker process per image (oracle/cpu_hot_path.py), every one set up before the common
    "go", the time taken until the last "done"; `cores` = processes x OpenMP threads, all of which work."""
    import bench

    c = bench.cpu_baseline(3, 32, 6, threads=2)  # 3 images on 2 processes (2 + 1), 32 x 32 rasters
    assert c["processes"] == 2 and c["omp_threads_per_process"] == 1 and c["cores"] == 2 and c["kind"] == "port"
    assert c["value"] > 0 and c["seconds_per_image_in_a_worker"] > 0 and len(c["sample"]) <= 200
    c = bench.cpu_baseline(1, 32, 1, threads=4)  # fewer images than threads: the rasteriser's OpenMP threads take the rest
    assert c["processes"] == 1 and c["omp_threads_per_process"] == 4 and c["cores"] == 4


def test_cpu_baseline_reports_the_knee_of_its_sweep(monkeypatch):
    """`cores` is the process count at which the figure stops improving (a doubling that buys less than 1.25 x ends the sweep
    and is NOT the reported level), not what the box claims to have: round 5's GPU boxes report 256 hardware threads and give
    a container 16.  A level that runs into its time limit ends the sweep with the best level so far; with no level done, the
    error surfaces."""
    import bench

    # a box whose quota is 16 cores: throughput doubles to 16 processes, stays flat beyond
    def fake(B_sample, is_, B_full, threads=None, limit_s=150.0):
        v = 0.01 * min(threads, 16) * (1.0 if threads <= 16 else 0.99)
        return {"value": v, "unit": "iters/s", "cores": threads, "kind": "port", "processes": threads, "omp_threads_per_process": 1,
                "seconds_per_image_in_a_worker": 1.0, "sample": f"{B_sample} images; {threads} processes x 1 OpenMP threads"}

    monkeypatch.setattr(bench, "cpu_baseline", fake)
    c = bench.cpu_baseline_at_the_knee(256, 64, 256)
    assert c["cores"] == 16 and [l["cores"] for l in c["sweep"]] == [8, 16, 32] and c["hardware_threads"] == 256
    assert c["at_8_threads"]["value"] == pytest.approx(0.08) and "knee" in c["sample"] and len(c["sample"]) <= 200
    c = bench.cpu_baseline_at_the_knee(256, 64, 256, sweep_all=True)  # (--cpu-sweep: every level, the knee still reported)
    assert c["cores"] == 16 and [l["cores"] for l in c["sweep"]] == [8, 16, 32, 64, 128, 256]
    assert bench.cpu_baseline_at_the_knee(64, 64, 4)["cores"] == 4  # (a box smaller than the first level)

    def starved(B_sample, is_, B_full, threads=None, limit_s=150.0):
        if threads > 8:
            raise TimeoutError("too slow")
        return fake(B_sample, is_, B_full, threads)

    monkeypatch.setattr(bench, "cpu_baseline", starved)
    assert bench.cpu_baseline_at_the_knee(256, 64, 256)["cores"] == 8
    monkeypatch.setattr(bench, "cpu_baseline", lambda *a, **k: (_ for _ in ()).throw(TimeoutError("nothing finished")))
    with pytest.raises(TimeoutError):
        bench.cpu_baseline_at_the_knee(256, 64, 256)
