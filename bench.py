#!/usr/bin/env python
"""bench.py -- trainmeshwarp optimiser-steps/sec with render + warp in the loop.

    python bench.py --gpus N --steps K --warmup W          # N > 1: starts its own N ranks (one per GPU, torchrun)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W   # ... or under a launcher: same ranks, same line

One "step" = one optimiser step of trainmeshwarp.py at B=64, 256x256 (BASELINE.json metric;
SURVEY Q15): a supervised data batch (B frames) + a consistency batch (B frame pairs): 3B
ResNet-18 forwards, MANO LBS, 2 differentiable renders (1780 verts / 7104 faces after
fill-back), occlusion check, 2-direction photometric pair loss, ONE backward through all of
it, Adam step.  Inputs are synthetic (seeded) and resident in HBM before the timed region.
Per-GPU work is fixed as N grows (weak scaling, batch-sharded DP; the model's gradients are
all-reduced over RCCL in 8 MB buckets issued from inside backward -- netscripts/gradreduce.py --
the render/warp kernels need no collective).

Rank 0 prints ONE JSON line.  Besides the contract fields it carries
  "roofline":     algorithmic bytes / live HIP-event duration of the dominant hot-path kernel,
  "kernels":      the same for every hot-path kernel group (render fwd, render bwd, pair loss...),
  "cpu_baseline": the CPU oracle (oracle/, kind "port") timed on this box's host cores on a
                  bounded sample of the same hot path (N=1 only).

Profiling switches (environment, scripts/ only; none of them is set in a measured run): HOC_KERNEL_GROUPS (group names
of --kernels-only, separated by ';'), HOC_FWD_DBG / HOC_BWD_FLAGS / HOC_FLOW_BWD_DBG (the kernels' own `flags >> 8` switches),
HOC_TILE_LIST, HOC_TILE_BOUND, HOC_GRAD_BOUND, HOC_PAIR_EMPTY (the warp kernels on coverage bytes that say "nothing rendered"),
HOC_TORCH_DDP / HOC_FORCE_DDP (A/B of the data-parallel path on one rank), HOC_TUNABLEOP.
"""
import argparse
import json
import os
import sys
import time

# this image's host driver supports only dmabuf IPC: without this RCCL's (and torch's) cross-process buffer
# sharing fails with "hipIpcGetMemHandle: invalid argument".  Read when the HIP runtime starts, i.e. it has to be
# in the environment before the first GPU call of the process; a value set by the launcher wins.
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np  # noqa: E402
import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=20)
    p.add_argument("--warmup", type=int, default=5)
    p.add_argument("--batch", type=int, default=64, help="per-GPU batch size (frames / frame pairs)")
    p.add_argument("--image-size", type=int, default=256, help="frame width (= side of the square raster)")
    p.add_argument("--image-height", type=int, default=None, help="frame height if not square (BASELINE config 3: 480 x 270)")
    p.add_argument("--encoder-dtype", choices=("f32", "bf16"), default="f32",
                   help="f32 = the reference's precision (headline); bf16 = BASELINE config 5: trunk under bf16 "
                        "autocast, heads / MANO / render / warp stay fp32 -- reported with dtype 'bf16+f32'")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--details-out", default=None, metavar="PATH",
                   help="where the FULL record goes (every kernel group, warp_tiles, in-step durations, notes): a JSON file, "
                        "default bench_details.json in the working directory; '-' = nowhere.  stdout carries the compact "
                        "contract line only (<= 4 KB)")
    p.add_argument("--no-kernel-bench", action="store_true")
    p.add_argument("--no-stock-trunk", action="store_true", help="skip the run with the stock trunk modules (the north-star-conformant figure)")
    p.add_argument("--stock-trunk-nchw", action="store_true", help="also time the stock modules in NCHW without solver search (rounds 1-4's leg)")
    p.add_argument("--cpu-sample", type=int, default=32, help="(unused since round 6: the sample is one image per worker process)")
    p.add_argument("--cpu-sweep", action="store_true", help="cpu_baseline also at 16 ... 256 worker processes (details file)")
    p.add_argument("--kernel-iters", type=int, default=50)
    p.add_argument("--kernels-only", action="store_true", help="only the per-kernel benchmark (profiling aid)")
    p.add_argument("--hot-only", action="store_true", help="only the render+warp hot path fwd+bwd (profiling aid)")
    p.add_argument("--roofline-only", action="store_true",
                   help="launch only the two roofline kernels a few times (what the in-run PMC passes profile)")
    p.add_argument("--no-pmc", action="store_true", help="skip the in-run rocprofv3 --pmc passes behind roofline.traffic")
    p.add_argument("--step-only", action="store_true",
                   help="only the timed training steps (no hot-path / stock-trunk / kernel / CPU legs): what the in-run "
                        "rocprofv3 --kernel-trace pass behind roofline.frac_in_step profiles")
    p.add_argument("--eager-step", action="store_true", help="(the only mode; kept for old command lines)")
    p.add_argument("--reducer-ab", type=int, default=0, metavar="PAIRS",
                   help="A/B inside ONE process (one model, one set of MIOpen / TunableOp solver choices): PAIRS x "
                        "(--steps plain steps, then --steps steps through the RCCL process group + bucketed gradient "
                        "reducer on one rank); prints {plain_ms, one_rank_rccl_ms, ratios}")
    return p.parse_args()


def event_time_ms(fn, iters, warmup=5, flush=None):
    """Average duration of fn() in ms, HIP events on torch's current stream (the stream every
    libmeshraster_hip launch of this process goes to).  With `flush` (a >= 512 MB tensor) each
    timed launch is preceded by a pass over that tensor so that the inputs are NOT resident in
    the 256 MB Infinity Cache / L2 (as in the training step, where each kernel runs once per
    render between unrelated work); only fn() is inside the event pair."""
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    if flush is None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters
    total = 0.0
    for _ in range(iters):
        flush.add_(1.0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        total += e0.elapsed_time(e1)
    return total / iters


# The raster backward AS THE TRAINING STEP LAUNCHES IT (round 4, ABI 5): the scatter of the pair loss's gradient to the vertex
# colours, the gradient itself -- for a unit coefficient, epilogue masks applied -- having been left by the FORWARD launch
# that held its taps in registers (mr_flow_pair_forward_grad_tiles -> mr_flow_pair_backward_unit_tiles).
# ROOF_BWD_R4A: the first round-4 form, one launch that recomputes the pair loss's backward in front of the scatter
# (mr_flow_pair_backward_tiles; what `roofline` described until this change).  ROOF_BWD_PLAIN: the scatter alone on a given
# flow gradient with the epilogue adjoint applied on the fly (mr_render_flow_backward, rounds 2-3).
ROOF_BWD = "flow_pair_backward_unit_tiles(train: E scatter of the forward's unit gradient x coefficient,2B)"
ROOF_BWD_R4A = "flow_pair_backward_tiles(train: pair-loss bwd + epilogue adjoint + E scatter,2B)"
ROOF_BWD_PLAIN = "render_flow_backward(train,E+epilogue adjoint,2B)"
ROOF_FWD = "render_flow_forward(train outputs,both frames=2B)"
FUSED_FWD = "flow_pair_forward_grad_tiles(train: occlusion + epilogue + pair loss + unit gradient, sparse)"
FUSED_FWD_PLAIN = "flow_pair_forward_tiles(occlusion + epilogue + pair loss, sparse)"
# the device kernels behind the groups (names as rocprofv3 prints them)
ROOF_KERNELS = {ROOF_BWD: ["unit_scatter_tiles_kernel"], ROOF_BWD_R4A: ["pair_scatter_tiles_kernel"],
                ROOF_BWD_PLAIN: ["scatter_tiles_kernel<true, true>"],
                ROOF_FWD: ["face_records_kernel<true>", "bin_boxes_kernel<false>", "raster_tile_kernel<true, true>"]}
# ... and as the training step launches the same render: the pair prologue has cleared the tile list's header, the per-face
# pass runs inside the binning kernel (MR_FLAG_TILE_LIST_CLEARED)
# (round 6: the binning launch of a pair step also runs the pair's vertex stage + stacked faces -- bin_boxes_prologue_kernel,
# which takes the place of pair_prologue_kernel + bin_boxes_kernel<true>; the in-step figure of the forward includes that work)
ROOF_KERNELS_IN_STEP = {ROOF_FWD: ["bin_boxes_prologue_kernel", "raster_tile_kernel<true, true>"]}
# (eight or more parts per image -- config 3's 16 renders: count launch + fill launch; MR_PAIR_STEP_SEPARATE_LAUNCHES: round 6's first form)
ROOF_KERNELS_IN_STEP_ALT = {ROOF_FWD: [["bin_count_prologue_kernel", "bin_fill_kernel<true>", "raster_tile_kernel<true, true>"],
                                       ["bin_boxes_kernel<true>", "raster_tile_kernel<true, true>"]]}
# compulsory bytes per pixel of a covered tile: face index 4 + vertex ids 12 + sampling weights 12 + ...
ROOF_BWD_PER_PIXEL = {ROOF_BWD: (36, "... + unit gradient 8"),
                      ROOF_BWD_R4A: (80, "... + three masks 12 + final flow 8 + source 12 + target 12 + two jitter values 8 (its scratch "
                                         "stays in the L2 of the workgroup that writes and re-reads it)"),
                      ROOF_BWD_PLAIN: (48, "... + flow gradient 8 + three masks 12")}
WARP_TILES_KERNELS_IN_STEP = {"flow_pair_forward_tiles_kernel<true>": "flow_pair_forward_tiles_kernel<true, true>"}
WARP_TILES_KERNELS = ("occlusion_flow_tiles_kernel", "pair_consist_forward_tiles_kernel", "pair_consist_backward_tiles_kernel",
                      "flow_pair_forward_tiles_kernel<false>", "flow_pair_forward_tiles_kernel<true>")


ROOF_OPTIONAL = ()
DEF_LANE_INSTS_PER_TERM = 10  # necessary lane-instructions per term of kernel D's walk (see kernel_bench)
# the warp half over the render's tile list (round 4) and what one pixel of a covered tile makes each pass move
WARP_TILES = ("occlusion_flow_tiles(train: occlusion + flow epilogue, sparse)", "pair_consist_forward_tiles(train, sparse)",
              "pair_consist_backward_tiles(train, sparse)", FUSED_FWD_PLAIN, FUSED_FWD)
WARP_TILES_BYTES = (44, 40, 48, 76, 84)
WARP_TILES_WHAT = ("per pixel of a covered tile: own mask 4 + scale 4 + flow 8, gathered flow 8 + scale 4 + mask 4, out occl 4 + flow 8",
                   "per pixel of a covered tile: flow 8 + source 12 + target 12 + two jitter values 8 (each image pixel counted once)",
                   "per pixel of a covered tile: flow 8 + source 12 + target 12 + two jitter values 8, out grad_flow 8",
                   "per pixel of a covered tile: the occlusion pass's 44 + source 12 + target 12 + two jitter values 8 (the flow it "
                   "warps with never leaves the thread)",
                   "per pixel of a covered tile: the occlusion pass's 44 + source 12 + target 12 + two jitter values 8 + out unit "
                   "gradient 8 (the step's forward launch when the vertices want a gradient)")


def kernel_bench(dev, B, is_, iters, only=None):
    """Each hot-path kernel group alone, on the bench workload's own tensors.  Algorithmic
    bytes per launch follow SURVEY 8(d) / DESIGN.md.  `only`: names of the groups to run."""
    from handobjectconsist_amd import _lib
    from handobjectconsist_amd.neurender import nr_ops
    from handobjectconsist_amd.utils import synth, textutils

    s = synth.random_scene(B, seed=0, image_size=is_)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    verts, faces_idx, K = t(s["verts1"]), t(s["faces"]), t(s["K1"])
    colors = torch.randn(B, verts.shape[1], 3, device=dev)
    tex = textutils.batch_vertex_textures(faces_idx, colors)
    faces_idx2 = torch.cat((faces_idx, faces_idx.flip(-1)), 1)
    tex2 = torch.cat((tex, tex.permute(0, 1, 4, 3, 2, 5)), 1).contiguous()
    v = nr_ops.projection(verts, K, torch.eye(3, device=dev)[None], torch.zeros(1, 3, device=dev),
                          torch.zeros(1, 5, device=dev), is_)
    faces = nr_ops.vertices_to_faces(v, faces_idx2).contiguous()
    F = faces.shape[1]
    npx = B * is_ * is_
    lib = _lib.load()
    st = _lib.stream_ptr(dev)
    f32 = dict(dtype=torch.float32, device=dev)
    rgb, alpha, depth = torch.empty((B, 3, is_, is_), **f32), torch.empty((B, is_, is_), **f32), torch.empty((B, is_, is_), **f32)
    fim = torch.empty((B, is_, is_), dtype=torch.int32, device=dev)
    wmap = torch.empty((B, is_, is_, 3), **f32)
    wbytes = int(lib.mr_render_workspace_bytes(B, F, is_))
    work = torch.empty((wbytes,), dtype=torch.uint8, device=dev)
    bg = torch.zeros(3, **f32)
    P = _lib.ptr

    def render_fwd():
        _lib.call("mr_render_forward", P(faces), P(tex2), P(bg), 0, P(rgb), P(alpha), P(depth), P(fim), P(wmap),
                  None, P(work), wbytes, B, F, is_, 2, 0.1, 100.0, 1e-3, 1, 1, 1, 0, st)

    g_rgb, g_alpha, g_depth = torch.randn_like(rgb), torch.randn_like(alpha), torch.randn_like(depth)
    grad_tex, grad_faces = torch.empty_like(tex2), torch.empty_like(faces)

    bl_bytes = int(lib.mr_render_backward_list_workspace_bytes(B, F))
    bl_work = torch.empty((bl_bytes,), dtype=torch.uint8, device=dev)

    def render_bwd_train():  # detach_renders=True: textures only (kernel E), as neurender.rasterize launches it
        _lib.call("mr_render_backward", P(faces), P(tex2), P(fim), P(rgb), P(alpha), P(g_rgb), None, None, None,
                  P(grad_tex), P(bl_work), bl_bytes, B, F, is_, 2, 0.1, 100.0, 1e-3, 1, 1, 1, int(os.environ.get("HOC_BWD_FLAGS", "0")), st)

    bw_bytes = int(lib.mr_render_backward_workspace_bytes(B, F, is_))
    bw_work = torch.empty((bw_bytes,), dtype=torch.uint8, device=dev)

    def render_bwd_full():  # kernels D + E + F
        _lib.call("mr_render_backward", P(faces), P(tex2), P(fim), P(rgb), P(alpha), P(g_rgb), P(g_alpha),
                  P(g_depth), P(grad_faces), P(grad_tex), P(bw_work), bw_bytes, B, F, is_, 2, 0.1, 100.0, 1e-3, 1, 1, 1,
                  int(os.environ.get("HOC_BWD_FLAGS", "0")), st)  # (flags >> 8: kernel D's profiling switches)

    # vertex-colour mode: what the training path actually launches (opticalflow -> render_vertex_colors)
    fidx32 = faces_idx.to(torch.int32).contiguous()
    v_c, g_cols = v.contiguous(), torch.empty_like(colors)
    F0 = fidx32.shape[1]

    def render_vc_fwd():
        _lib.call("mr_render_vc_forward", P(v_c), P(fidx32), P(colors), P(bg), 0, P(rgb), P(alpha), P(depth), P(fim),
                  P(wmap), P(work), wbytes, B, v_c.shape[1], F0, 1, is_, 0.1, 100.0, 1e-3, 1, 1, 1, 0, 0, st)

    def render_vc_bwd():
        _lib.call("mr_render_vc_backward", P(v_c), P(fidx32), P(fim), P(wmap), P(depth), P(g_rgb), P(g_cols), B, v_c.shape[1], F0, 1, is_,
                  1e-3, 0, 0, st)

    # ... and AS the training step launches it: get_opticalflow renders frame 1 and frame 2 of every pair
    # in one launch over 2B meshes (warping/opticalflow.py), forward and backward
    B2 = 2 * B
    v2_c = nr_ops.projection(t(s["verts2"]), t(s["K2"]), torch.eye(3, device=dev)[None], torch.zeros(1, 3, device=dev),
                             torch.zeros(1, 5, device=dev), is_)
    pv = torch.cat([v_c, v2_c], 0).contiguous()
    pf = torch.cat([fidx32, fidx32], 0).contiguous()
    pcols, pg_cols = torch.randn(B2, pv.shape[1], 3, device=dev), torch.empty(B2, pv.shape[1], 3, device=dev)
    prgb, palpha, pdepth = torch.empty((B2, 3, is_, is_), **f32), torch.empty((B2, is_, is_), **f32), torch.empty((B2, is_, is_), **f32)
    pfim, pwmap = torch.empty((B2, is_, is_), dtype=torch.int32, device=dev), torch.empty((B2, is_, is_, 3), **f32)
    pwbytes = int(lib.mr_render_workspace_bytes(B2, F, is_))
    pwork = torch.empty((pwbytes,), dtype=torch.uint8, device=dev)
    pg_rgb = torch.randn_like(prgb)

    def render_vc_fwd_pair():
        _lib.call("mr_render_vc_forward", P(pv), P(pf), P(pcols), P(bg), 0, P(prgb), P(palpha), P(pdepth), P(pfim),
                  P(pwmap), P(pwork), pwbytes, B2, pv.shape[1], F0, 1, is_, 0.1, 100.0, 1e-3, 1, 1, 1, 0, 0, st)

    def render_vc_bwd_pair():
        _lib.call("mr_render_vc_backward", P(pv), P(pf), P(pfim), P(pwmap), P(pdepth), P(pg_rgb), P(pg_cols), B2,
                  pv.shape[1], F0, 1, is_, 1e-3, 0, 0, st)

    pmask = torch.empty((B2, is_, is_), **f32)
    keep_lut = torch.ones(2 * F0 + 2, **f32)
    keep_lut[torch.tensor(synth.HAND_IGNORE_FACES, device=dev) + 1] = 0

    ptile_hit = torch.empty((B2, (is_ + 7) // 8, (is_ + 31) // 32, 4), dtype=torch.uint8, device=dev)
    # per-pixel records of the flow-mode render: vertex ids + sampling weights (its own weight buffer: pwmap holds the
    # barycentrics of the full-output render)
    pvid = torch.empty((B2, is_, is_, 3), dtype=torch.int32, device=dev)
    pwrec = torch.empty((B2, is_, is_, 3), **f32)
    poccl = (torch.rand((B2, is_, is_), device=dev) < 0.9).float()
    pg_flow = torch.randn((B2, is_, is_, 2), **f32)
    pg_bound = pg_flow.abs().amax((1, 2, 3)).contiguous()  # (in training: left by the pair loss's backward kernel)

    # the list length of the previous launch, written by the kernel into pinned host memory: the next launch's grid
    tile_word = torch.zeros(1, dtype=torch.int32).pin_memory()
    fwd_dbg = int(os.environ.get("HOC_FWD_DBG", "0")) << 8         # profiling experiments (raster_fwd.hip)
    fwd_list = os.environ.get("HOC_TILE_LIST", "1") == "1"

    def render_flow_fwd_pair():  # the training path's output set: rgb planes 0 / 1, alpha, flow mask, face index
        last = int(tile_word[0])
        bound = (last + last // 8 + 64 if last > 0 else -1) if fwd_list else 0
        _lib.call("mr_render_flow_forward", P(pv), P(pf), P(pcols), P(bg), 0, P(keep_lut), int(keep_lut.numel()), 0.99999,
                  P(prgb), P(palpha), P(pmask), None, P(pwrec), P(pfim), P(ptile_hit), P(pwork), pwbytes, B2, pv.shape[1], F0, 1, is_, 0.1, 100.0, 1e-3,
                  _lib.FLAG_SPARSE_TILES | fwd_dbg, P(pvid), bound, P(tile_word), P(pg_cols), int(pg_cols.numel()), 0, st)

    def render_vc_bwd_pair_recompute():  # ... and its backward: no weight / depth maps to read back
        _lib.call("mr_render_vc_backward", P(pv), P(pf), P(pfim), None, None, P(pg_rgb), P(pg_cols), B2, pv.shape[1], F0, 1,
                  is_, 1e-3, 0, 0, st)

    def render_flow_bwd_pair():  # what the training step launches: flow-space gradient + epilogue masks in, d colours out
        # (as in the step: the output was cleared by the forward's binning pass -- here once, the repeated launches of the
        # timing loop keep adding into it, which changes no instruction the kernel executes)
        _lib.call("mr_render_flow_backward", P(pv), P(pf), P(pfim), P(ptile_hit), P(pwrec), None, None, P(pg_flow),
                  P(pmask), P(pmask[:B]), P(palpha[B:]), B, P(poccl), is_, is_, P(pg_cols), B2, pv.shape[1], F0, 1, is_, 1e-3,
                  _lib.FLAG_OUTPUT_ZEROED | (int(os.environ.get("HOC_FLOW_BWD_DBG", "0")) << 8), P(pvid), 0,
                  P(pg_bound) if os.environ.get("HOC_GRAD_BOUND", "1") == "1" else None, st)

    render_flow_fwd_pair()
    render_vc_fwd_pair()
    im_ref, im, jm_ref, jm = [t(a) for a in synth.random_images(B, is_, is_, 0)]
    # the flows the pair loss sees in training: rendered displacement fields (zero outside the meshes, a few pixels
    # inside), from the flow-mode render + occlusion / epilogue pass of this very scene
    pflows = torch.empty((B2, is_, is_, 2), **f32)
    pcols_flow = (pcols * 1.5).contiguous()
    _lib.call("mr_render_flow_forward", P(pv), P(pf), P(pcols_flow), P(bg), 0, P(keep_lut), int(keep_lut.numel()), 0.99999,
              P(prgb), P(palpha), P(pmask), None, P(pwrec), P(pfim), P(ptile_hit), P(pwork), pwbytes, B2, pv.shape[1],
              F0, 1, is_, 0.1, 100.0, 1e-3, _lib.FLAG_SPARSE_TILES, P(pvid), -1, P(tile_word), None, 0, 0, st)
    torch.cuda.synchronize()
    pocc = torch.empty((B2, is_, is_), **f32)
    # the render's tile list (in its workspace): what the sparse warp kernels of the training path are launched over
    tlist = _lib.tile_list(pwork, B2, F, is_)
    tl_bound = int(tile_word[0]) + int(tile_word[0]) // 8 + 64
    if os.environ.get("HOC_TILE_BOUND"):  # profiling: workgroups of the listed launches (fewer than the list: grid-stride rounds)
        tl_bound = int(os.environ["HOC_TILE_BOUND"])

    def occlusion_flow():  # occlusion check + flow epilogue of both directions (what the training step launches)
        _lib.call("mr_occlusion_flow", P(pmask[:B]), P(palpha[B:]), P(prgb[:B]), P(prgb[B:]), 3 * is_ * is_, P(pmask[:B]),
                  P(pmask[B:]), P(pocc[:B]), P(pocc[B:]), P(pflows[:B]), P(pflows[B:]), P(ptile_hit[:B]), P(ptile_hit[B:]), B, is_,
                  is_, is_, is_, 0.03, 0.99999, st)

    def occlusion_flow_tiles():  # ... as the training step launches it since round 4: over the tile list, sparse outputs
        _lib.call("mr_occlusion_flow_tiles", P(pmask[:B]), P(palpha[B:]), P(prgb[:B]), P(prgb[B:]), 3 * is_ * is_, P(pmask[:B]),
                  P(pmask[B:]), P(pocc[:B]), P(pocc[B:]), P(pflows[:B]), P(pflows[B:]), P(ptile_hit[:B]), P(ptile_hit[B:]), B, is_,
                  is_, is_, 0.03, 0.99999, tlist[0], tlist[1], tlist[2], tl_bound, st)

    occlusion_flow()
    if os.environ.get("HOC_ZERO_FLOWS"):  # experiment: the pair kernels on all-zero flows (nothing but the flow reads)
        pflows.zero_()
    flow12, flow21 = pflows[:B], pflows[B:]
    pbytes = int(lib.mr_pair_consist_workspace_bytes(B, is_, is_))
    pcwork = torch.empty((pbytes,), dtype=torch.uint8, device=dev)
    sums, lf, lb = torch.empty((B, 4), **f32), torch.empty((B,), **f32), torch.empty((B,), **f32)
    g12, g21 = torch.empty_like(flow12), torch.empty_like(flow21)
    gl = torch.full((B,), 1.0 / B, **f32)

    def pair_fwd():
        _lib.call("mr_pair_consist_forward", P(flow12), P(flow21), P(im_ref), P(im), P(jm_ref), P(jm), 3, P(pcwork),
                  pbytes, P(sums), P(lf), P(lb), *([None] * 8), B, is_, is_, 0.99999, P(ptile_hit[:B]), P(ptile_hit[B:]), is_, st)

    pgmax = torch.zeros(2 * B, **f32)  # per-image gradient maxima, as the training path asks for them

    def pair_bwd():
        _lib.call("mr_pair_consist_backward", P(flow12), P(flow21), P(im_ref), P(im), P(jm_ref), P(jm), 3, P(sums),
                  P(gl), P(gl), P(g12), P(g21), B, is_, is_, 0.99999, P(ptile_hit[:B]), P(ptile_hit[B:]), is_,
                  P(pgmax) if os.environ.get("HOC_GRAD_BOUND", "1") == "1" else None, st)

    ptbytes = int(lib.mr_pair_consist_tiles_workspace_bytes(B, is_))
    ptwork = torch.empty((ptbytes,), dtype=torch.uint8, device=dev)

    def pair_fwd_tiles():
        _lib.call("mr_pair_consist_forward_tiles", P(flow12), P(flow21), P(im_ref), P(im), P(jm_ref), P(jm), 3, P(ptwork), ptbytes,
                  P(sums), P(lf), P(lb), B, is_, is_, 0.99999, P(ptile_hit[:B]), P(ptile_hit[B:]), is_, tlist[0], tlist[1],
                  tlist[2], tl_bound, st)

    def pair_bwd_tiles():
        _lib.call("mr_pair_consist_backward_tiles", P(flow12), P(flow21), P(im_ref), P(im), P(jm_ref), P(jm), 3, P(sums), P(gl),
                  P(gl), P(g12), P(g21), B, is_, is_, 0.99999, P(ptile_hit[:B]), P(ptile_hit[B:]), is_,
                  P(pgmax) if os.environ.get("HOC_GRAD_BOUND", "1") == "1" else None, tlist[0], tlist[1], tlist[2], tl_bound, st)

    def flow_pair_fwd_tiles():  # the step's forward launch: occlusion + epilogue + pair loss, one pass over the list
        _lib.call("mr_flow_pair_forward_tiles", P(pmask[:B]), P(palpha[B:]), P(prgb[:B]), P(prgb[B:]), 3 * is_ * is_, P(pmask[:B]),
                  P(pmask[B:]), P(pocc[:B]), P(pocc[B:]), P(pflows[:B]), P(pflows[B:]), P(ptile_hit[:B]), P(ptile_hit[B:]),
                  P(im_ref), P(im), P(jm_ref), P(jm), 3, P(ptwork), ptbytes, P(sums), P(lf), P(lb), B, is_, is_, is_, 0.03, 0.99999,
                  0.99999, tlist[0], tlist[1], tlist[2], tl_bound, st)

    punit, punit_max, lsum = torch.empty((B2, is_, is_, 2), **f32), torch.empty((B2,), **f32), torch.empty((B,), **f32)
    pswork = torch.empty((int(_lib.load().mr_flow_pair_scatter_work_bytes(B, is_)),), dtype=torch.uint8, device=dev)

    def flow_pair_fwd_grad_tiles():  # ... and as the step launches it when the vertices want a gradient (they do)
        _lib.call("mr_flow_pair_forward_grad_tiles", P(pmask[:B]), P(palpha[B:]), P(prgb[:B]), P(prgb[B:]), 3 * is_ * is_,
                  P(pmask[:B]), P(pmask[B:]), P(pocc[:B]), P(pocc[B:]), P(pflows[:B]), P(pflows[B:]), P(ptile_hit[:B]),
                  P(ptile_hit[B:]), P(im_ref), P(im), P(jm_ref), P(jm), 3, P(ptwork), ptbytes, P(sums), P(lf), P(lb), B, is_, is_,
                  is_, 0.03, 0.99999, 0.99999, tlist[0], tlist[1], tlist[2], tl_bound, P(punit), P(punit_max), P(lsum), P(pswork), st)

    def flow_pair_bwd_unit_tiles():  # the step's backward launch (output cleared by the forward's binning pass, as in the step)
        _lib.call("mr_flow_pair_backward_unit_tiles", P(pfim), P(ptile_hit), P(pwrec), P(pvid), P(punit), P(punit_max), P(sums),
                  P(gl), P(gl), is_, is_, P(pg_cols), B2, pv.shape[1], F0, 1, is_, 1e-3,
                  _lib.FLAG_OUTPUT_ZEROED | (int(os.environ.get("HOC_FLOW_BWD_DBG", "0")) << 8), 0, P(pswork), st)

    pscratch = torch.empty((B2, is_, is_, 2), **f32)

    def flow_pair_bwd_tiles():  # the step's backward launch (output cleared by the forward's binning pass, as in the step)
        _lib.call("mr_flow_pair_backward_tiles", P(pfim), P(ptile_hit), P(pwrec), P(pvid), P(pflows), P(im_ref), P(im), P(jm_ref),
                  P(jm), 3, P(sums), P(gl), P(gl), P(pmask), P(pmask[:B]), P(palpha[B:]), P(pocc), P(pscratch), is_, is_,
                  P(pg_cols), B2, pv.shape[1], F0, 1, is_, 1e-3, 0.99999,
                  _lib.FLAG_OUTPUT_ZEROED | (int(os.environ.get("HOC_FLOW_BWD_DBG", "0")) << 8), 0, st)

    m1, m2 = alpha.unsqueeze(1).contiguous(), alpha.unsqueeze(1).contiguous()
    o1, o2 = torch.empty((B, is_, is_), **f32), torch.empty((B, is_, is_), **f32)

    def occlusion():
        _lib.call("mr_occlusion_mask", P(m1), P(m2), P(rgb), P(rgb), 3 * is_ * is_, None, None, P(o1), P(o2), B, is_,
                  is_, 0.03, 0.99999, st)

    # dataset side: the 3B frames of a step, 640x480 decoded frames -> is x is inputs + jitter masks
    from handobjectconsist_amd.datasets import handutils

    NF, Hs_, Ws_ = 3 * B, 480, 640
    rngf = np.random.default_rng(0)
    fr_u8 = torch.randint(0, 256, (NF, Hs_, Ws_, 3), dtype=torch.uint8, device=dev)
    fr_coeffs = torch.from_numpy(np.stack([handutils.pil_coeffs(handutils.get_affine_transform(
        rngf.uniform((250, 180), (390, 300)), rngf.uniform(250, 420), (is_, is_))[0]) for _ in range(NF)])).to(dev)
    fr_img, fr_mask = torch.empty((NF, 3, is_, is_), **f32), torch.empty((NF, 3, is_, is_), **f32)
    fr_wb = int(lib.mr_frames_to_batch_workspace_bytes(NF, is_, is_))
    fr_work = torch.empty((fr_wb,), dtype=torch.uint8, device=dev)

    def frames_to_batch():
        _lib.call("mr_frames_to_batch", P(fr_u8), P(fr_coeffs), None, 0.5, 0.5, 0.5, 1.0, 1.0, 1.0, P(fr_work), fr_wb,
                  P(fr_img), P(fr_mask), 3, NF, Hs_, Ws_, is_, is_, st)

    # trainer side: what sits between MIOpen's convolutions in the ResNet-18 trunk (channels-last
    # kernels, the layout the trunk runs in; the buffers are plain memory of the right size), at the shapes of a
    # step (3B frames): the stem (bn + relu + max-pool on [3B,64,is/2,is/2]) and a layer-1 bn + identity + relu
    hs = is_ // 2
    st_x = torch.randn(NF, 64, hs, hs, **f32)
    st_y, st_gx = torch.empty(NF, 64, hs // 2, hs // 2, **f32), torch.empty_like(st_x)
    st_gy = torch.randn_like(st_y)
    st_am = torch.empty(st_y.shape, dtype=torch.uint8, device=dev)  # arg-max positions (channels-last kernels)
    bnp = [torch.rand(64, **f32) + 0.5, torch.randn(64, **f32), torch.randn(64, **f32), torch.rand(64, **f32) + 0.5]
    bn_gw, bn_gb = torch.empty(64, **f32), torch.empty(64, **f32)
    st_wb = int(lib.mr_stem_pool_backward_workspace_bytes(NF, 64, hs, hs))
    st_work = torch.empty((st_wb,), dtype=torch.uint8, device=dev)
    l1_x, l1_res, l1_gy = st_gx[:, :, :hs // 2, :hs // 2].contiguous(), torch.randn_like(st_y), torch.randn_like(st_y)
    l1_y, l1_gx, l1_gr = torch.empty_like(st_y), torch.empty_like(st_y), torch.empty_like(st_y)
    l1_wb = int(lib.mr_bn_act_backward_workspace_bytes(NF, 64))
    l1_work = torch.empty((l1_wb,), dtype=torch.uint8, device=dev)
    l1_x.normal_()

    def stem_fwd():
        _lib.call("mr_stem_pool_forward", P(st_x), *[P(t_) for t_ in bnp], 1e-5, 0, 1, P(st_y), P(st_am), NF, 64, hs, hs, st)

    def stem_bwd():
        _lib.call("mr_stem_pool_backward", P(st_gy), None, P(st_x), P(st_am), *[P(t_) for t_ in bnp], 1e-5, 0, 1, P(st_gx), P(bn_gw),
                  P(bn_gb),
                  P(st_work), st_wb, NF, 64, hs, hs, st)

    def bn_fwd():
        _lib.call("mr_bn_act_forward", P(l1_x), P(l1_res), *[P(t_) for t_ in bnp], 1e-5, 1, 0, 1, P(l1_y), NF, 64,
                  (hs // 2) ** 2, st)

    def bn_bwd():
        _lib.call("mr_bn_act_backward", P(l1_gy), None, P(l1_x), P(l1_res), *[P(t_) for t_ in bnp], 1e-5, 1, 0, 1, P(l1_gx), P(l1_gr),
                  P(bn_gw), P(bn_gb), P(l1_work), l1_wb, NF, 64, (hs // 2) ** 2, st)

    render_fwd()
    BF = B * F
    groups = [
        # name, fn, algorithmic bytes per launch
        ("render_forward", render_fwd, 132 * BF + 36 * npx),
        ("render_vc_forward(train)", render_vc_fwd, 132 * BF + 36 * npx),
        ("render_backward_train(E)", render_bwd_train, (12 + 4 + 12 + 4) * npx + (36 + 96) * BF),
        ("render_vc_backward(train,E)", render_vc_bwd, (12 + 4 + 12 + 4) * npx + (36 + 96) * BF),
        ("render_vc_forward(train,both frames=2B)", render_vc_fwd_pair, 2 * (132 * BF + 36 * npx)),
        ("render_vc_backward(train,E,both frames=2B)", render_vc_bwd_pair, 2 * ((12 + 4 + 12 + 4) * npx + (36 + 96) * BF)),
        ("render_flow_forward(train outputs,both frames=2B)", render_flow_fwd_pair, 2 * (132 * BF + 36 * npx)),
        ("render_vc_backward(train,E,2B,recomputed weights)", render_vc_bwd_pair_recompute, 2 * ((12 + 4 + 12 + 4) * npx + (36 + 96) * BF)),
        # kernel E of SURVEY 8(d) (same algorithmic bytes) + the adjoint of the flow epilogue folded in
        ("render_flow_backward(train,E+epilogue adjoint,2B)", render_flow_bwd_pair, 2 * ((12 + 4 + 12 + 4) * npx + (36 + 96) * BF)),
        ("render_backward_full(D+E+F)", render_bwd_full, 56 * npx + 168 * BF),
        ("pair_consist_forward", pair_fwd, 48 * npx),
        ("pair_consist_backward", pair_bwd, 64 * npx),
        # the same three passes as the training step launches them (round 4): over the render's tile list, outputs
        # written under the covered tiles only -- same SURVEY 8(d) bytes, `compulsory_bytes` = what they have to move
        (WARP_TILES[0], occlusion_flow_tiles, (8 + 16 + 8 + 16) * npx),
        (WARP_TILES[1], pair_fwd_tiles, 48 * npx),
        (WARP_TILES[2], pair_bwd_tiles, 64 * npx),
        # ... and fused, as the training step launches them now: SURVEY 8(d)'s bytes of the passes each replaces
        (FUSED_FWD_PLAIN, flow_pair_fwd_tiles, (8 + 16 + 8 + 16) * npx + 48 * npx),
        (ROOF_BWD_R4A, flow_pair_bwd_tiles, 64 * npx + 2 * ((12 + 4 + 12 + 4) * npx + (36 + 96) * BF)),
        # ... and with the pair loss's backward moved into the forward launch: the pair forward + the pair backward's bytes
        # there, kernel E's here
        (FUSED_FWD, flow_pair_fwd_grad_tiles, (8 + 16 + 8 + 16) * npx + 48 * npx + 64 * npx),
        (ROOF_BWD, flow_pair_bwd_unit_tiles, 2 * ((12 + 4 + 12 + 4) * npx + (36 + 96) * BF)),
        ("occlusion_mask", occlusion, (8 + 16 + 8) * npx),
        # + the two final flows written in the same pass (16 B per pixel)
        ("occlusion_flow(train: occlusion + flow epilogue)", occlusion_flow, (8 + 16 + 8 + 16) * npx),
        # 3 source bytes in, 12 B image + 12 B three-channel jitter mask out, per output pixel of the 3B frames
        ("frames_to_batch(3B frames,640x480->crop)", frames_to_batch, (3 + 12 + 12) * NF * is_ * is_),
        # encoder glue: bytes = tensors read + written once
        ("stem_bn_relu_maxpool_forward[3B,64]", stem_fwd, 4 * (st_x.numel() + st_y.numel())),
        ("stem_bn_relu_maxpool_backward[3B,64]", stem_bwd, 4 * (2 * st_x.numel() + st_y.numel())),
        ("bn_add_relu_forward(layer1)[3B,64]", bn_fwd, 4 * 3 * l1_x.numel()),
        ("bn_add_relu_backward(layer1)[3B,64]", bn_bwd, 4 * 5 * l1_x.numel()),
    ]
    out = {}
    if only is not None:
        groups = [g for g in groups if g[0] in only]
    if os.environ.get("HOC_PAIR_EMPTY") == "1":  # profiling aid: the warp kernels on coverage bytes that say "nothing rendered"
        ptile_hit.zero_()
    flush = torch.zeros(768 * 1024 * 1024 // 4, **f32)  # 768 MB > Infinity Cache (256 MB)
    if any(g[0] == ROOF_BWD for g in groups):
        flow_pair_fwd_grad_tiles()  # (the unit gradient that launch scatters, whatever subset of the groups runs)
    for name, fn, nbytes in groups:
        ms = event_time_ms(fn, iters, flush=flush)
        ms_warm = event_time_ms(fn, iters)
        gbs = nbytes / (ms * 1e-3) / 1e9
        out[name] = {"ms": round(ms, 4), "ms_cache_warm": round(ms_warm, 4), "algorithmic_MB": round(nbytes / 1e6, 1), "algorithmic_bytes": int(nbytes),
                     "GBps": round(gbs, 1), "frac_hbm_peak": round(gbs / HBM_PEAK_GBS, 4),
                     "frac_hbm_peak_cache_warm": round(nbytes / (ms_warm * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
    del flush
    full = "render_backward_full(D+E+F)"
    if full in out:
        # D + E + F is bound by vector-instruction issue, not by HBM (DESIGN.md sections 11 / 12).  The bound is ALGORITHMIC: the
        # number of terms of upstream's walk on this scene (counted by the strip kernel itself in one extra, untimed launch with
        # the profiling switch flags >> 8 & 1024; include/meshraster_hip.h: mr_pixel_map_terms) x the lane-instructions a term
        # cannot do without -- the difference-times-gradient sum over four channels 5, the distance 1, its reciprocal 3
        # (v_rcp_f32 is a quarter-rate instruction), the accumulation 1 --
        # on 1024 SIMDs that issue one wave-instruction (64 lanes) per 4 cycles at 2.4 GHz.  (Round 5 divided by the kernel's
        # OWN measured SQ_INSTS_VALU: a kernel with twice the necessary instructions scored the same.)
        import ctypes

        word = ctypes.c_uint64(0)
        lib.mr_pixel_map_terms(ctypes.byref(word), 1)
        prev = os.environ.get("HOC_BWD_FLAGS")
        os.environ["HOC_BWD_FLAGS"] = str(int(prev or "0") | (1024 << 8))
        try:
            render_bwd_full()
        finally:
            if prev is None:
                del os.environ["HOC_BWD_FLAGS"]
            else:
                os.environ["HOC_BWD_FLAGS"] = prev
        lib.mr_pixel_map_terms(ctypes.byref(word), 1)
        terms = int(word.value)
        if terms > 0:
            issue_ms = terms * DEF_LANE_INSTS_PER_TERM / 64 * 4 / (1024 * 2.4e9) * 1e3
            out[full].update({"terms": terms, "lane_insts_per_term": DEF_LANE_INSTS_PER_TERM,
                              "lane_insts_per_term_are": "difference x gradient over 4 channels 5, distance 1, reciprocal 3, accumulate 1",
                              "algorithmic_issue_ms": round(issue_ms, 4),
                              "frac_of_algorithmic_issue": round(issue_ms / out[full]["ms"], 4),
                              "frac_of_algorithmic_issue_cache_warm": round(issue_ms / out[full]["ms_cache_warm"], 4)})
    covered_words = int((ptile_hit.view(torch.int32) != 0).sum())
    for name, per_px, what in zip(WARP_TILES, WARP_TILES_BYTES, WARP_TILES_WHAT):
        if name in out:
            comp = covered_words * 32 * 8 * per_px + ptile_hit.numel()
            k = out[name]
            k.update({"covered_tiles": covered_words, "tiles": int(ptile_hit.numel() // 4), "compulsory_bytes": int(comp),
                      "compulsory_bytes_are": what,
                      "compulsory_GBps": round(comp / (k["ms"] * 1e-3) / 1e9, 1),
                      "compulsory_frac_hbm_peak": round(comp / (k["ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                      "compulsory_frac_hbm_peak_cache_warm": round(comp / (k["ms_cache_warm"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)})
    # What the raster backward of the training path HAS to move, at least once, for this scene: per tile the forward reports
    # as covered (coverage bytes) the 256 pixels of ROOF_BWD_PER_PIXEL's bytes, the coverage bytes themselves and the
    # [2B,V,3] output.  The SURVEY 8(d) figure (`algorithmic_bytes`: 32 B for every pixel of the raster + 132 B per face)
    # describes upstream's kernel E, which these kernels replace without touching a [B,F,...] tensor or the 83 % of the
    # screen that is empty: dividing THAT by the launch time gives an effective rate that can exceed the chip's bandwidth
    # and is kept only as `frac_algorithmic`.
    for name, (per_px, _what) in ROOF_BWD_PER_PIXEL.items():
        if name in out:
            comp = covered_words * 32 * 8 * per_px + ptile_hit.numel() + pg_cols.numel() * 4
            k = out[name]
            k.update({"covered_tiles": covered_words, "tiles": int(ptile_hit.numel() // 4), "compulsory_bytes": int(comp),
                      "compulsory_per_pixel": per_px,
                      "compulsory_GBps": round(comp / (k["ms"] * 1e-3) / 1e9, 1),
                      "compulsory_frac_hbm_peak": round(comp / (k["ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                      "compulsory_frac_hbm_peak_cache_warm": round(comp / (k["ms_cache_warm"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)})
    return out


def cpu_baseline(B_sample, is_, B_full, threads=None, limit_s=150.0):
    """The CPU oracle (oracle/, a port -- the reference has no CPU render path) on a bounded sample of the hot
    path: 2 renders + flow masks + occlusion + pair loss forward, and the texture / flow backward; extrapolated linearly in
    the batch size (images are independent).  One worker PROCESS per image (oracle/cpu_hot_path.py; round 5's thread pool
    serialised on the GIL: 256 threads bought 1.06 x over 8): `threads` processes of one image each while the sample allows
    (`B_sample` caps it; beyond that the rasteriser's OpenMP threads take the rest), all started and set up before a
    common "go", timed until the last one is done.  `cores` = processes x OpenMP threads per process."""
    import subprocess

    threads = threads or os.cpu_count() or 1
    n_tasks = max(1, min(B_sample, threads))
    omp = max(1, threads // n_tasks)
    counts = [B_sample * (i + 1) // n_tasks - B_sample * i // n_tasks for i in range(n_tasks)]
    env = dict(os.environ, OMP_NUM_THREADS=str(omp), PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    procs = [subprocess.Popen([sys.executable, "-m", "oracle.cpu_hot_path", "--seed", str(i), "--count", str(c), "--size", str(is_),
                               "--omp", str(omp)], cwd=ROOT, env=env, stdin=subprocess.PIPE, stdout=subprocess.PIPE, text=True)
             for i, c in enumerate(counts)]
    try:
        for p_ in procs:
            if p_.stdout.readline().strip() != "ready":
                raise RuntimeError("cpu_baseline: a worker process did not come up (oracle/cpu_hot_path.py)")
        t0 = time.perf_counter()
        for p_ in procs:
            p_.stdin.write("go\n")
            p_.stdin.flush()
        own = []
        import select

        for p_ in procs:
            # (a level that does not finish within `limit_s` is abandoned: a starved box must not cost the run its line)
            left = limit_s - (time.perf_counter() - t0)
            if left <= 0 or not select.select([p_.stdout], [], [], left)[0]:
                raise TimeoutError(f"cpu_baseline: {n_tasks} worker processes did not finish within {limit_s:.0f} s")
            word = p_.stdout.readline().split()
            if len(word) != 2 or word[0] != "done":
                raise RuntimeError("cpu_baseline: a worker process failed (oracle/cpu_hot_path.py)")
            own.append(float(word[1]))
        dt = time.perf_counter() - t0
    finally:
        for p_ in procs:
            if p_.poll() is None:
                p_.kill()
            p_.wait()
    sec_per_iter = dt * (B_full / B_sample)
    return {"value": round(1.0 / sec_per_iter, 6), "unit": "iters/s", "cores": n_tasks * omp, "kind": "port",
            "processes": n_tasks, "omp_threads_per_process": omp, "seconds_per_image_in_a_worker": round(sum(own) / len(own) / max(counts), 3),
            "sample": f"render+warp hot path fwd+bwd (no encoder), {B_sample} images of B={B_full} at {is_}x{is_}, {dt:.1f} s x{B_full / B_sample:g}; "
                      f"{n_tasks} processes x {omp} OpenMP threads"}


def cpu_baseline_at_the_knee(is_, B_full, ncpu, sweep_all=False, gain=1.25, budget_s=60.0):
    """`cpu_baseline` at 8, 16, 32, ... worker processes (one image each) up to the box's hardware threads, stopping at the first
    doubling that buys less than `gain` (or when `budget_s` of CPU-leg wall time is spent): the figure reported is the BEST level's
    and `cores` is that level's process count -- the count at which the figure stops improving, not `os.cpu_count()` (round 5's
    GPU boxes report 256 hardware threads and give a container a fraction of them: 256 workers ran 33 x slower PER IMAGE than 8).
    Every level measured rides along in `sweep`; `at_8_threads` lines up with BASELINE.md's 8-thread reference measurement."""
    levels, t0 = [], time.perf_counter()
    t_ = min(8, ncpu)
    best = None
    while True:
        try:
            c_ = cpu_baseline(max(1, min(t_, 4 * B_full)), is_, B_full, threads=t_)
        except TimeoutError as e:
            sys.stderr.write(f"[bench] {e}\n")
            if best is None:
                raise
            break
        levels.append({"cores": c_["cores"], "value": c_["value"], "s_per_image": c_["seconds_per_image_in_a_worker"]})
        improved = best is None or c_["value"] >= gain * best["value"]
        if improved:  # (a level that buys less than `gain` over the best so far is past the knee: the best stays)
            best = c_
        if t_ >= ncpu or (not improved and not sweep_all) or time.perf_counter() - t0 > budget_s:
            break
        t_ = min(2 * t_, ncpu)
    out = dict(best)
    out["hardware_threads"] = ncpu
    out["sweep"] = levels
    out["sample"] += f"; cores = knee of sweep {[l['cores'] for l in levels]} ({ncpu} hw threads reported)"
    if levels[0]["cores"] == min(8, ncpu):
        out["at_8_threads"] = {"value": levels[0]["value"]}
    return out


def pmc_traffic_in_run(args, timeout=240):
    """HBM bytes per launch of the roofline kernels, measured NOW: two `rocprofv3 --pmc` passes (FETCH_SIZE, then
    WRITE_SIZE -- they do not fit one pass, MI355X_MICROARCH.md PMC slots) over `bench.py --roofline-only` as a
    subprocess, mean per dispatch and kernel.  FETCH_SIZE / WRITE_SIZE count KB; on gfx950 FETCH_SIZE reports
    half of the bytes of wide coalesced reads (same guide, HBM section), so `hbm_bytes` = 1024 (2 FETCH + WRITE) is
    the upper estimate and `hbm_bytes_low` = 1024 (FETCH + WRITE) the lower one.  {} when rocprofv3 is missing or a
    pass fails (the committed profiles/ then hold the last good numbers)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile

    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return {}
    sums = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        tmp = tempfile.mkdtemp(prefix="hoc_pmc_", dir="/tmp")
        cmd = [exe, "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", tmp, "-o", "p", "--", sys.executable,
               os.path.abspath(__file__), "--roofline-only", "--batch", str(args.batch), "--image-size", str(args.image_size),
               "--kernel-iters", "3"]
        try:
            subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), timeout=timeout, capture_output=True, check=True)
            files = glob.glob(os.path.join(tmp, "**", "p_counter_collection.csv"), recursive=True)
            per = {}
            with open(files[0]) as fh:
                for row in csv.DictReader(fh):
                    if row["Counter_Name"] != counter:
                        continue
                    key = (row["Kernel_Name"].split("(")[0].replace("void ", "").replace("mr::", ""), row["Dispatch_Id"])
                    per[key] = per.get(key, 0.0) + float(row["Counter_Value"])
            for (kname, _), v in per.items():
                sums.setdefault(kname, {}).setdefault(counter, []).append(v)
        except Exception as e:  # noqa: BLE001 -- the benchmark line must not depend on the profiler
            sys.stderr.write(f"[bench] PMC pass {counter} failed: {e}\n")
            return {}
        finally:
            shutil.rmtree(tmp, ignore_errors=True)
    out = {}
    for kname, c in sums.items():
        if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            fe, wr = sum(c["FETCH_SIZE"]) / len(c["FETCH_SIZE"]), sum(c["WRITE_SIZE"]) / len(c["WRITE_SIZE"])
            out[kname] = {"FETCH_SIZE_KB": round(fe, 1), "WRITE_SIZE_KB": round(wr, 1), "hbm_bytes": int(1024 * (2 * fe + wr)),
                          "hbm_bytes_low": int(1024 * (fe + wr)), "dispatches": len(c["FETCH_SIZE"])}
    return out


def in_step_durations(args, timeout=600):
    """Durations of this build's kernels INSIDE training steps: one `rocprofv3 --kernel-trace` pass over
    `bench.py --step-only` (a few steps of the same workload, nothing else in the process), median per kernel name in
    microseconds.  MIOpen's measured solver search and TunableOp are off in that subprocess (it only has to reach the
    steady state quickly; the render / warp kernels do not depend on them).  {} when rocprofv3 is missing or the pass fails."""
    import csv
    import glob
    import shutil
    import statistics
    import subprocess
    import tempfile

    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return {}
    tmp = tempfile.mkdtemp(prefix="hoc_step_", dir="/tmp")
    cmd = [exe, "--kernel-trace", "--output-format", "csv", "-d", tmp, "-o", "p", "--", sys.executable, os.path.abspath(__file__),
           "--step-only", "--eager-step", "--steps", "6", "--warmup", "3", "--batch", str(args.batch), "--image-size",
           str(args.image_size)]  # (eager: the kernels are the replayed step's, and every profiler version lists them)
    if args.image_height:
        cmd += ["--image-height", str(args.image_height)]
    try:
        subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp", HOC_CUDNN_BENCHMARK="0", HOC_TUNABLEOP="0"),
                       timeout=timeout, capture_output=True, check=True)
        files = glob.glob(os.path.join(tmp, "**", "p_kernel_trace.csv"), recursive=True)
        per = {}
        with open(files[0]) as fh:
            for row in csv.DictReader(fh):
                name = row["Kernel_Name"]
                if "mr::" not in name:
                    continue
                key = name.split("(")[0].replace("void ", "").replace("mr::", "")
                per.setdefault(key, []).append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
        return {k: {"median_us": round(statistics.median(v), 2), "min_us": round(min(v), 2), "dispatches": len(v)} for k, v in per.items()}
    except Exception as e:  # noqa: BLE001 -- informational
        sys.stderr.write(f"[bench] in-step kernel-trace pass failed: {e}\n")
        return {}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def roofline_block(name, k, pmc, units, in_step=None):
    """`frac` = bytes / HIP-event duration with cold caches / peak, where bytes = the kernel's own COMPULSORY traffic when
    the kernel bench states it (the raster backward: covered tiles only -- cannot exceed 1 by construction) and SURVEY
    8(d)'s algorithmic bytes otherwise; `frac_algorithmic` always carries the SURVEY 8(d) figure; `dram_frac` = the bytes
    the PMC counters saw / the same duration / peak -- what the DRAM interface actually carried."""
    own = "compulsory_bytes" in k
    roof = {"kernel": name, "bound": "hbm", "achieved": k["compulsory_GBps"] if own else k["GBps"], "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": k["compulsory_frac_hbm_peak"] if own else k["frac_hbm_peak"],
            "frac_cache_warm": k["compulsory_frac_hbm_peak_cache_warm"] if own else k["frac_hbm_peak_cache_warm"],
            "bytes": k["compulsory_bytes"] if own else k["algorithmic_bytes"],
            "bytes_are": ("compulsory traffic of this launch: %d B per pixel of the %d covered tiles of %d (face index 4 + vertex ids "
                          "12 + sampling weights 12 %s) + coverage bytes + output"
                          % (k["compulsory_per_pixel"], k["covered_tiles"], k["tiles"], ROOF_BWD_PER_PIXEL[name][1])) if own
                         else "algorithmic bytes of SURVEY 8(d)",
            "traffic": None, "algorithmic_bytes": k["algorithmic_bytes"], "achieved_algorithmic": k["GBps"],
            "frac_algorithmic": k["frac_hbm_peak"], "frac_algorithmic_cache_warm": k["frac_hbm_peak_cache_warm"],
            "launch_ms": k["ms"], "launch_ms_cache_warm": k["ms_cache_warm"],
            "units_per_launch": units, "device_kernels": ROOF_KERNELS[name]}
    step_names = ROOF_KERNELS_IN_STEP.get(name, ROOF_KERNELS[name])
    for alt in ROOF_KERNELS_IN_STEP_ALT.get(name, []) + [ROOF_KERNELS[name]]:
        if in_step and not all(d in in_step for d in step_names):
            step_names = alt
    if in_step and all(d in in_step for d in step_names):
        us = sum(in_step[d]["median_us"] for d in step_names)
        roof.update({"in_step_us": round(us, 2), "frac_in_step": round(roof["bytes"] / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
                     "in_step_kernels": step_names,
                     "in_step_source": "median duration of the kernel(s) inside training steps, rocprofv3 --kernel-trace pass "
                                       "over `bench.py --step-only` run by this process"})
    found = [pmc[d] for d in ROOF_KERNELS[name] if d in pmc]
    if len(found) >= len([d for d in ROOF_KERNELS[name] if d not in ROOF_OPTIONAL]):
        hi, lo = sum(f["hbm_bytes"] for f in found), sum(f["hbm_bytes_low"] for f in found)
        roof.update({"traffic": hi, "traffic_low": lo, "traffic_unit": "bytes/launch",
                     "traffic_source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes run by this process "
                                       "(1024 x (2 FETCH + WRITE); traffic_low: FETCH as reported)",
                     "dram_frac": round(hi / (k["ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                     "dram_frac_low": round(lo / (k["ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)})
    return roof


CONTRACT_LINE_MAX = 4096  # bytes: the driver's record keeps the tail of stdout and parses its LAST line (round 5's 21 KB line was lost)


def _pick(d, keys):
    return None if d is None else {k: d.get(k) for k in keys if k in d}


def _short(text, n=200):
    text = str(text)
    return text if len(text) <= n else text[: n - 3] + "..."


def contract_line(full):
    """The ONE line stdout carries: the bench contract's fields + the `roofline` and `cpu_baseline` objects with the figures
    a reader needs, no prose beyond <= 200-character strings, <= CONTRACT_LINE_MAX bytes whatever the run produced
    (optional keys are dropped from the back until it fits).  Everything else -- every kernel group, the warp passes, the
    in-step durations, notes -- is the details file (`--details-out`)."""
    roof, fwd, kernels, cpu = full.get("roofline"), full.get("roofline_forward"), full.get("kernels") or {}, full.get("cpu_baseline")
    line = {k: full.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                                     "scaling", "vs_baseline", "dtype", "data", "config", "headline", "step_mode", "hot_path_ms")}
    line["unit"] = _short(line["unit"])
    line["step_mode"] = _short(line["step_mode"], 80)
    line["config"] = dict(line["config"], workload=_short(line["config"]["workload"], 260))
    r = None
    if roof is not None:
        r = _pick(roof, ("kernel", "bound", "achieved", "peak", "unit", "frac", "frac_cache_warm", "frac_in_step", "in_step_us",
                         "bytes", "traffic", "traffic_low", "dram_frac", "algorithmic_bytes", "frac_algorithmic", "launch_ms",
                         "launch_ms_cache_warm", "device_kernels"))
        r["kernel"] = _short(r["kernel"], 120)
        r["forward"] = _pick(fwd, ("frac", "frac_cache_warm", "frac_in_step", "launch_ms", "launch_ms_cache_warm", "in_step_us",
                                   "bytes", "traffic", "device_kernels"))
        full_k = kernels.get("render_backward_full(D+E+F)")
        if full_k is not None:
            r["d_e_f"] = {"ms": full_k["ms"], "ms_cache_warm": full_k["ms_cache_warm"], "frac": full_k["frac_hbm_peak"],
                          "bytes": full_k["algorithmic_bytes"]}
            for k in ("frac_of_algorithmic_issue", "algorithmic_issue_ms", "terms"):
                if k in full_k:
                    r["d_e_f"][k] = full_k[k]
        fused = (full.get("warp_tiles") or {}).get(FUSED_FWD)
        if fused is not None:
            r["fused_warp_forward"] = _pick(fused, ("frac", "frac_cache_warm", "frac_in_step", "launch_ms", "in_step_us", "bytes"))
        for k in ("hot_path_device_ms", "hot_path_eager_ms_host_bound", "stock_trunk_it_s"):
            r[k] = roof.get(k)
    line["roofline"] = r
    c = None
    if cpu is not None:
        c = _pick(cpu, ("value", "unit", "cores", "kind", "sample"))
        c["sample"] = _short(c["sample"])
        if "at_8_threads" in cpu:
            c["at_8_threads"] = {"value": cpu["at_8_threads"]["value"]}
    line["cpu_baseline"] = c
    ranks = full.get("ranks")
    if ranks is not None:
        line["ranks"] = {"backend": ranks["backend"], "world_size": ranks["world_size"], "reducer": _short(ranks["reducer"], 100),
                         "grad_allreduce_MB": ranks["grad_allreduce_MB"], "bucket_MB": ranks["bucket_MB"],
                         "ms_per_step_by_rank": [r_["ms_per_step"] for r_ in sorted(ranks["per_rank"], key=lambda r_: r_["rank"])],
                         "device_by_rank": [r_["device"] for r_ in sorted(ranks["per_rank"], key=lambda r_: r_["rank"])]}
    line["details"] = full.get("details")
    # never more than the driver reads: shed optional keys, least important first
    shed = [("roofline", "fused_warp_forward"), ("roofline", "device_kernels"), ("ranks", "device_by_rank"), ("details",),
            ("roofline", "forward", "device_kernels"), ("step_mode",), ("ranks", "reducer"), ("roofline", "d_e_f"), ("ranks",)]
    while len(json.dumps(line)) > CONTRACT_LINE_MAX and shed:
        path = shed.pop(0)
        d = line
        for k in path[:-1]:
            d = d.get(k) if isinstance(d, dict) else None
        if isinstance(d, dict):
            d.pop(path[-1], None)
    return line


def self_launch(args):
    """``python bench.py --gpus N`` started as a PLAIN script (no RANK / WORLD_SIZE in the environment) with N > 1:
    become the launcher -- re-execute this very command line under ``python -m torch.distributed.run --nnodes=1
    --nproc-per-node N --master-addr 127.0.0.1 --master-port <free port>``, one rank per GPU.  The ranks inherit this
    process' stdout, so rank 0's ONE JSON line is this command's output.  Never returns."""
    import socket

    share = os.environ.get("HOC_SHARE_GPU", "0") == "1"
    have = torch.cuda.device_count()
    if have < 1:
        sys.exit("bench.py needs a GPU")
    if have < args.gpus and not share:
        sys.exit(f"bench.py: --gpus {args.gpus} asks for {args.gpus} devices, this node shows {have} "
                 f"(HOC_SHARE_GPU=1 HOC_DIST_BACKEND=gloo lets the ranks share devices: tests only)")
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stderr.write("[bench] --gpus %d without a launcher: %s\n" % (args.gpus, " ".join(cmd)))
    sys.stderr.flush()
    # one OpenMP thread per rank unless the caller says otherwise (what torchrun itself would set, without its warning)
    os.environ.setdefault("OMP_NUM_THREADS", "1")
    os.execv(sys.executable, cmd)


_T0 = time.perf_counter()


def _phase(name):
    """Wall-clock note on stderr (rank 0): how long each leg of a default run takes."""
    if os.environ.get("RANK", "0") == "0":
        sys.stderr.write(f"[bench] {time.perf_counter() - _T0:7.1f} s  {name}\n")
        sys.stderr.flush()


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ:
        self_launch(args)
    # stdout carries exactly ONE line, the JSON record: everything else that writes to file descriptor 1 (RCCL prints its
    # version banner there, buffered until exit) is sent to stderr for the duration of the run
    real_stdout = os.dup(1)
    sys.stdout.flush()
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    assert torch.cuda.is_available(), "bench.py needs a GPU"
    # MIOpen's measured solver search (first call of every convolution shape, inside the warm-up steps) instead of its
    # find-db heuristics: 27.4 instead of 28.0 ms per step with the channels-last trunk; HOC_CUDNN_BENCHMARK=0 disables
    torch.backends.cudnn.benchmark = os.environ.get("HOC_CUDNN_BENCHMARK", "1") == "1"
    # ... and PyTorch's TunableOp for the fp32 GEMMs of the heads (rocBLAS' default pick for [192,512] x [512,256] is a
    # single-workgroup kernel: 91 us for 25 MFLOP, twice per step): every GEMM shape is timed over the rocBLAS / hipBLASLt
    # solutions at its first call, inside the warm-up steps; 26.8 instead of 27.2 ms per step.  HOC_TUNABLEOP=0 disables.
    # Its results file goes to the temp directory, not the working tree.
    if os.environ.get("HOC_TUNABLEOP", "1") == "1" and not (args.kernels_only or args.roofline_only or args.hot_only):
        import tempfile
        torch.cuda.tunable.enable(True)
        torch.cuda.tunable.tuning_enable(True)
        torch.cuda.tunable.set_max_tuning_duration(10)    # ms per candidate solution
        torch.cuda.tunable.set_max_tuning_iterations(20)
        torch.cuda.tunable.set_filename(os.path.join(tempfile.gettempdir(), "hoc_tunableop_%d.csv" % os.getpid()))
    # HOC_SHARE_GPU=1 (tests only): ranks beyond the device count share GPUs, with HOC_DIST_BACKEND=gloo -- RCCL
    # refuses two ranks on one device; this is how a world_size-2 job is exercised on the one-GPU test box
    dev_index = local_rank % torch.cuda.device_count() if os.environ.get("HOC_SHARE_GPU", "0") == "1" else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    dist = None
    # HOC_FORCE_DDP=1: take the multi-GPU code path (RCCL process group, bucketed gradient reducer, barriers, max-over-ranks
    # all-reduce) with a single rank too -- how the path is exercised on a one-GPU box (tests/test_gpu_bench.py)
    use_dist = world > 1 or os.environ.get("HOC_FORCE_DDP", "0") == "1" or args.reducer_ab > 0
    if use_dist:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        backend = os.environ.get("HOC_DIST_BACKEND", "nccl")  # "nccl" is RCCL on ROCm
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks "
                 f"(plain `python bench.py --gpus N` starts its own N ranks)")

    from handobjectconsist_amd import _lib
    from handobjectconsist_amd.models.synthnet import SynthMeshRegNet
    from handobjectconsist_amd.models.warpreg import WarpRegNet
    from handobjectconsist_amd.netscripts.epochpassconsist import SyntheticConsistLoader, raise_pending_nan, train_step

    assert _lib.load().mr_device_ok() == 1, "libmeshraster_hip.so: no gfx950 device"
    if args.kernels_only or args.roofline_only:
        only = (ROOF_BWD, ROOF_BWD_R4A, ROOF_BWD_PLAIN, ROOF_FWD) + WARP_TILES if args.roofline_only else None
        if os.environ.get("HOC_KERNEL_GROUPS"):  # profiling aid: group names of kernel_bench, separated by ";"
            only = tuple(os.environ["HOC_KERNEL_GROUPS"].split(";"))
        os.write(real_stdout, (json.dumps(kernel_bench(dev, args.batch, args.image_size, args.kernel_iters, only), indent=1) + "\n").encode())
        return
    torch.manual_seed(rank)
    B, is_ = args.batch, args.image_size
    model = SynthMeshRegNet().to(dev)
    model.eval()  # --freeze_batchnorm: BN statistics frozen, affine parameters trainable
    if args.encoder_dtype == "bf16":
        model.encoder_dtype = torch.bfloat16
    net, reducer = model, None
    torch_ddp = os.environ.get("HOC_TORCH_DDP", "0") == "1"  # A/B only: torch's DistributedDataParallel wrapper
    if use_dist and torch_ddp:
        from torch.nn.parallel import DistributedDataParallel as DDP

        net = DDP(model, device_ids=[dev_index], bucket_cap_mb=int(os.environ.get("HOC_DDP_BUCKET_MB", "16")),
                  broadcast_buffers=False, gradient_as_bucket_view=os.environ.get("HOC_DDP_BUCKET_VIEW", "1") == "1")
    elif use_dist and not args.reducer_ab:
        from handobjectconsist_amd.netscripts.gradreduce import BucketedGradReducer

        # the model is NOT wrapped: 8 MB buckets, filled by one multi-tensor copy each from inside backward(), one
        # asynchronous RCCL all-reduce (average) per bucket overlapping the encoder backward; frozen BN statistics
        # -> no buffer exchange (netscripts/gradreduce.py)
        reducer = BucketedGradReducer(model.parameters(), bucket_mb=int(os.environ.get("HOC_DDP_BUCKET_MB", "16")))
    ih_ = args.image_height or is_
    assert ih_ <= is_, "--image-height must not exceed --image-size (the raster is the square of the longer side)"
    premodel = WarpRegNet((is_, ih_), net, lambda_consist=0.001, lambda_data=0.999, criterion="l1", gt_refs=True,
                          progressive_steps=1000, use_backward=True, mano_faces=model.mano_layer.th_faces,
                          pair_outputs="loss").to(dev)
    premodel.step_count = 1000  # past the lambda ramp: the consistency term carries its full weight
    # trainmeshwarp.py's optimiser (Adam, lr 5e-5).  fused=True is stock PyTorch's single-pass kernel for the
    # same update (A/B on one MI355X: 46.05 -> 45.10 ms per step); HOC_FUSED_ADAM=0 selects the foreach default
    params = [p for p in model.parameters() if p.requires_grad]
    # The step is issued eagerly.  (A hipGraph replay of the whole step was built in round 5, found to return garbage convolution
    # weight gradients now and then, and lives in scripts/graph_step_experiment.py since round 6 -- not in the product.)
    fused_adam = os.environ.get("HOC_FUSED_ADAM", "1") == "1"
    optimizer = torch.optim.Adam(params, lr=5e-5, fused=fused_adam)
    loader = SyntheticConsistLoader(B, is_, seed=rank, device=dev, pool=2, image_height=ih_)

    def barrier():
        if dist is not None:
            dist.barrier()

    # the reference raises on a NaN loss before backward / step (epochpassconsist.py:61-63): one host sync per step,
    # kept inside the timed region (HOC_CHECK_NAN=0 measures the step without it)
    check_nan = os.environ.get("HOC_CHECK_NAN", "1") == "1"
    if args.reducer_ab:
        # the cost of the data-parallel code path before a byte is communicated, decided inside one process: the plain
        # loop and the reducer loop share the model, the optimiser state and every solver / GEMM choice
        from handobjectconsist_amd.netscripts.gradreduce import BucketedGradReducer

        def block(red):
            for i in range(args.warmup):
                train_step(loader.step_batches(i), premodel, optimizer, check_nan=check_nan, reducer=red)
            torch.cuda.synchronize()
            t_ = time.perf_counter()
            for i in range(args.steps):
                train_step(loader.step_batches(i), premodel, optimizer, check_nan=check_nan, reducer=red)
            if check_nan:
                raise_pending_nan(optimizer)
            torch.cuda.synchronize()
            return (time.perf_counter() - t_) / args.steps * 1e3

        block(None)  # solver searches + allocator warm-up, outside every measured block
        plain_ms, red_ms, nb = [], [], 0
        for _ in range(args.reducer_ab):
            plain_ms.append(round(block(None), 4))
            red = BucketedGradReducer(model.parameters(), bucket_mb=int(os.environ.get("HOC_DDP_BUCKET_MB", "16")))
            nb = len(red.buckets)
            red_ms.append(round(block(red), 4))
            red.remove()
            optimizer.zero_grad(set_to_none=True)
            del red
        out = {"plain_ms": plain_ms, "one_rank_rccl_ms": red_ms, "ratios": [round(b_ / a_, 4) for a_, b_ in zip(plain_ms, red_ms)],
               "backend": "rccl" if dist.get_backend() == "nccl" else dist.get_backend(), "buckets": nb, "steps": args.steps,
               "warmup": args.warmup, "what": "one process, one model; alternating blocks without / with the bucketed gradient "
                                              "reducer (one rank: no byte leaves the device)"}
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
        dist.destroy_process_group()
        return
    _phase("model built; warm-up steps")

    def make_step(pre, opt):
        return lambda batches: train_step(batches, pre, opt, check_nan=check_nan, reducer=reducer)

    step_fn = make_step(premodel, optimizer)
    n_warm = args.warmup
    for i in range(0 if args.hot_only else n_warm):
        step_fn(loader.step_batches(i))
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    loss = torch.zeros(1)
    for i in range(0 if args.hot_only else args.steps):
        loss, _ = step_fn(loader.step_batches(n_warm + i))
    if check_nan:
        raise_pending_nan(optimizer)  # the last step's device-side NaN flag (train_step's contract), inside the timed region
    torch.cuda.synchronize()
    t_local = time.perf_counter() - t0  # this rank's own K steps, before it waits for the others
    barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ranks = None
    if dist is not None:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
        # evidence that N ranks ran: every rank's device, its own step time and the collective backend
        mine = torch.tensor([float(rank), float(dev_index), t_local / max(args.steps, 1) * 1e3], device=dev, dtype=torch.float64)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        ranks = {"backend": "rccl" if dist.get_backend() == "nccl" else dist.get_backend(), "world_size": world,
                 "per_rank": [{"rank": int(r_[0]), "device": int(r_[1]), "ms_per_step": round(float(r_[2]), 3)} for r_ in allr],
                 "grad_allreduce_MB": round(sum(p_.numel() for p_ in model.parameters() if p_.requires_grad) * 4 / 1e6, 1),
                 "bucket_MB": int(os.environ.get("HOC_DDP_BUCKET_MB", "16")),
                 "reducer": "torch DistributedDataParallel (HOC_TORCH_DDP=1)" if torch_ddp else
                            f"gradreduce.BucketedGradReducer, {len(reducer.buckets)} buckets, all-reduce issued from backward"}
    assert torch.isfinite(loss).all(), "loss is not finite"
    if dist is not None and os.environ.get("HOC_CHECK_REPLICAS", "0") == "1":
        # data-parallel replicas must stay bit-identical: same averaged gradients, same Adam update on every rank
        chk = torch.stack([p_.detach().double().sum() for p_ in model.parameters()]).sum().reshape(1)
        allchk = [torch.zeros_like(chk) for _ in range(world)]
        dist.all_gather(allchk, chk)
        assert all(torch.equal(c_, allchk[0]) for c_ in allchk), f"replicas diverged: {[float(c_) for c_ in allchk]}"

    # hot path alone (2 renders, flows, occlusion, pair loss, backward to the vertices)
    hot_ms = None
    if args.step_only:
        if rank == 0:
            os.write(real_stdout, (json.dumps({"ms_per_step": round(dt / args.steps * 1e3, 3), "steps": args.steps}) + "\n").encode())
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return
    if rank == 0:
        from handobjectconsist_amd.models import warpbranch

        consist = loader.step_batches(0)[1]
        fake_results = [{"recov_handverts3d": s_["_handverts3d"].clone().requires_grad_(True),
                         "recov_objverts3d": s_["_objverts3d"].clone().requires_grad_(True)}
                        for s_ in consist["data"]]

        hot_leaves = [v for r_ in fake_results for v in r_.values()]

        def hot():
            l, _ = warpbranch.forward(consist["data"], fake_results, premodel.th_faces, premodel.renderer, (is_, ih_),
                                      premodel.criterion, gt_refs=True, hand_ignore_faces=premodel.hand_ignore_faces,
                                      use_backward=True, pair_outputs="loss")
            # (the gradients w.r.t. the vertices are RETURNED, as the step hands them on to the MANO layer's backward;
            # `l.backward()` would add five device-to-device copies into the leaves' .grad, 12 us that no step contains)
            return torch.autograd.grad(l, hot_leaves, allow_unused=True)

        _phase("timed steps done; hot path")
        hot_ms = event_time_ms(hot, 10, 3)
        # The eager loop above is bound by the HOST once the kernels are this short (~0.58 ms of Python + launch calls per
        # pass against ~0.33 ms of device work; inside a training step the host issues this section while the device is
        # still busy with the encoder, so there it is the device time that counts).  Device time: the same pass captured
        # once into a hipGraph and replayed back to back.
        hot_eager_ms, hot_graph_ms = hot_ms, None
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(2):
                    hot()
            torch.cuda.current_stream().wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            # (captured on the stream of the passes above: the pair step's plan -- keyed by stream -- is the warm one)
            with torch.cuda.graph(graph, stream=side):
                hot()
            hot_graph_ms = event_time_ms(graph.replay, 20, 3)
            hot_ms = hot_graph_ms
        except Exception as e:  # noqa: BLE001 -- informational figure; the eager one stays
            sys.stderr.write(f"[bench] hot path could not be captured into a graph: {e}\n")
            torch.cuda.synchronize()
        if args.hot_only:
            os.write(real_stdout, (json.dumps({"hot_path_ms": hot_ms}) + "\n").encode())
            return

    # The north-star-CONFORMANT figure ("the ResNet-18 encoder and regression heads stay on stock PyTorch-ROCm"): the same
    # step with the trunk's BatchNorm2d / ReLU / MaxPool2d as the stock nn modules.  It differs from `value` in ONE thing --
    # the module substitution; memory format (channels-last), MIOpen's measured solver search (cudnn.benchmark) and TunableOp
    # are stock PyTorch features and stay as `value` has them.  `--stock-trunk-nchw` adds the round-1..4 leg (NCHW, solver
    # search off) as stock_trunk_nchw.  Single GPU only.
    _phase("hot path done; stock trunk")
    stock, stock_nchw = None, None
    if rank == 0 and world == 1 and not use_dist and not args.hot_only and not args.no_stock_trunk \
            and args.encoder_dtype == "f32":
        from handobjectconsist_amd.models import synthnet as _sn

        def stock_leg(channels_last, solver_search, what):
            saved = (_sn.USE_HIP_BN, _sn.USE_CHANNELS_LAST, torch.backends.cudnn.benchmark)
            _sn.USE_HIP_BN, _sn.USE_CHANNELS_LAST, torch.backends.cudnn.benchmark = False, channels_last, solver_search
            try:
                torch.manual_seed(rank)
                model_s = SynthMeshRegNet().to(dev)
                model_s.eval()
                pre_s = WarpRegNet((is_, ih_), model_s, lambda_consist=0.001, lambda_data=0.999, criterion="l1", gt_refs=True,
                                   progressive_steps=1000, use_backward=True, mano_faces=model_s.mano_layer.th_faces,
                                   pair_outputs="loss").to(dev)
                pre_s.step_count = 1000
                opt_s = torch.optim.Adam([p for p in model_s.parameters() if p.requires_grad], lr=5e-5, fused=fused_adam)
                step_s = make_step(pre_s, opt_s)
                for i in range(max(n_warm, 2)):
                    step_s(loader.step_batches(i))
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for i in range(args.steps):
                    step_s(loader.step_batches(max(n_warm, 2) + i))
                if check_nan:
                    raise_pending_nan(opt_s)
                torch.cuda.synchronize()
                dt_s = time.perf_counter() - t1
                del model_s, pre_s, opt_s, step_s
                return {"what": what, "ms_per_step": round(dt_s / args.steps * 1e3, 3), "value": round(args.steps / dt_s, 4)}
            finally:
                _sn.USE_HIP_BN, _sn.USE_CHANNELS_LAST, torch.backends.cudnn.benchmark = saved
                torch.cuda.empty_cache()

        stock = stock_leg(_sn.USE_CHANNELS_LAST, torch.backends.cudnn.benchmark,
                          "same step, same settings (channels-last, cudnn.benchmark and TunableOp as in `value`), trunk on the STOCK "
                          "PyTorch-ROCm modules (nn.BatchNorm2d / ReLU / MaxPool2d): differs from `value` in the module substitution "
                          "only; render + warp + MANO + heads unchanged")
        if args.stock_trunk_nchw:
            stock_nchw = stock_leg(False, False, "stock modules, NCHW, cudnn.benchmark off (the stock_trunk leg of rounds 1-4)")

    kernels, roof, roof_fwd, cpu, warp_tiles, in_step_line = None, None, None, None, None, None
    if rank == 0 and not args.no_kernel_bench:
        _phase("stock trunk done; kernel bench")
        # (N > 1: the other ranks sit in the final barrier meanwhile -- the roofline groups only, a few launches each)
        kernels = kernel_bench(dev, B, is_, args.kernel_iters) if world == 1 else \
            kernel_bench(dev, B, is_, min(args.kernel_iters, 10), (ROOF_BWD, ROOF_BWD_R4A, ROOF_BWD_PLAIN, ROOF_FWD) + WARP_TILES)
        _phase("kernel bench done; PMC passes")
        pmc = {} if (args.no_pmc or world > 1) else pmc_traffic_in_run(args)
        _phase("PMC passes done; in-step kernel trace")
        in_step = {} if (args.no_pmc or world > 1) else in_step_durations(args)
        _phase("in-step trace done")
        units = f"{2 * B} renders of {is_}x{is_}, 7104 faces"
        # the raster backward (north star) in the shape the training step launches it: one launch for both frames of
        # the pair, the pair loss's backward and the adjoint of the flow epilogue folded in
        roof = roofline_block(ROOF_BWD, kernels[ROOF_BWD], pmc, units, in_step)
        roof["note"] = ("the pair loss's backward runs inside the step's FORWARD launch since ABI 5 (warp_tiles: " + FUSED_FWD + "); "
                        "this launch is the scatter to the vertex colours alone.  `recomputing_form`: the launch it replaced "
                        "(pair-loss backward recomputed in front of the scatter), `scatter_alone`: the round-3 launch on a given "
                        "flow gradient")
        roof["recomputing_form"] = roofline_block(ROOF_BWD_R4A, kernels[ROOF_BWD_R4A], pmc, units)
        roof["scatter_alone"] = roofline_block(ROOF_BWD_PLAIN, kernels[ROOF_BWD_PLAIN], pmc, units)
        # ... and the forward of the same launch shape: the hot-path kernel that takes the most time
        roof_fwd = roofline_block(ROOF_FWD, kernels[ROOF_FWD], pmc, units, in_step)
        # the warp half of the step (launched over the render's tile list): compulsory-bytes fractions + PMC traffic
        warp_tiles = {}
        for name, dk in zip(WARP_TILES, WARP_TILES_KERNELS):
            k = kernels[name]
            w = {"kernel": dk, "launch_ms": k["ms"], "launch_ms_cache_warm": k["ms_cache_warm"], "bytes": k["compulsory_bytes"],
                 "bytes_are": k["compulsory_bytes_are"], "frac": k["compulsory_frac_hbm_peak"],
                 "frac_cache_warm": k["compulsory_frac_hbm_peak_cache_warm"], "frac_algorithmic": k["frac_hbm_peak"],
                 "traffic": None}
            if dk in pmc:
                w.update({"traffic": pmc[dk]["hbm_bytes"], "traffic_low": pmc[dk]["hbm_bytes_low"],
                          "write_bytes": int(pmc[dk]["WRITE_SIZE_KB"] * 1024), "fetch_bytes_as_reported": int(pmc[dk]["FETCH_SIZE_KB"] * 1024)})
            # (inside a pair step the fused warp forward reads the render's 16-byte pixel records: the <true, true> instantiation)
            sk = WARP_TILES_KERNELS_IN_STEP.get(dk, dk) if WARP_TILES_KERNELS_IN_STEP.get(dk, dk) in in_step else dk
            if sk in in_step:
                w.update({"in_step_us": in_step[sk]["median_us"], "in_step_kernel": sk,
                          "frac_in_step": round(w["bytes"] / (in_step[sk]["median_us"] * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)})
            warp_tiles[name] = w
        in_step_line = in_step or None
        # what the driver's record keeps of this line is `roofline` / `cpu_baseline`: the figures a reader needs ride inside
        roof["forward"] = {k: roof_fwd.get(k) for k in ("kernel", "device_kernels", "launch_ms", "launch_ms_cache_warm", "bytes",
                                                       "bytes_are", "frac", "frac_cache_warm", "in_step_us", "frac_in_step",
                                                       "in_step_kernels", "traffic")}
        roof["hot_path_device_ms"] = None if hot_graph_ms is None else round(hot_graph_ms, 3)
        roof["hot_path_eager_ms_host_bound"] = None if hot_ms is None else round(hot_eager_ms, 3)
        roof["stock_trunk_it_s"] = None if stock is None else stock["value"]
        roof["stock_trunk_is"] = None if stock is None else "north-star-conformant step (stock BatchNorm2d / ReLU / MaxPool2d modules; everything else as `value`)"
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        _phase("CPU baseline")
        ncpu = os.cpu_count() or 1
        try:
            cpu = cpu_baseline_at_the_knee(is_, B, ncpu, sweep_all=args.cpu_sweep)
        except (TimeoutError, RuntimeError, OSError) as e:  # (the line must not depend on the host's CPUs being available)
            sys.stderr.write(f"[bench] cpu_baseline failed: {e}\n")
            cpu = None

    if rank == 0:
        ms = dt / args.steps * 1e3
        line = {
            "metric": f"trainmeshwarp iters/sec (render+warp, B={B}, {is_}x{ih_})",
            "headline": bool(B == 64 and is_ == 256 and ih_ == 256 and args.encoder_dtype == "f32"),  # BASELINE.json's metric config
            "value": round(world * args.steps / dt, 4),
            "unit": "iters/s (one optimiser step = 1 data batch + 1 consist batch of B per GPU; roofline.stock_trunk_it_s = the same "
                    "step with the trunk's BatchNorm/ReLU/max-pool on the stock nn modules)",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms, 3),
            "step_mode": "eager launches",
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if args.encoder_dtype == "f32" else "bf16 encoder + f32 render/warp (BASELINE config 5, not the headline)",
            "data": "synthetic",
            "config": {"workload": f"trainmeshwarp.py consist step, per-GPU B={B}, {is_}x{ih_}, hand 778v/1552f + "
                                   f"object 1002v/2000f (7104 faces after fill-back), ResNet-18 {('fp32' if args.encoder_dtype == 'f32' else 'bf16-autocast') + ' (MIOpen convolutions, channels-last, + fused HIP BatchNorm/ReLU/residual/max-pool kernels)'}, Adam",
                       "global_batch": B * world, "image_size": is_, "parallelism": f"dp{world}"},
            "hot_path_ms": None if hot_ms is None else round(hot_ms, 3),
            # (what the scalar above is, so that lines of different rounds are not compared blindly: rounds 1-2 reported the
            # eager, host-bound figure under this key; roofline.frac is on the kernel's compulsory bytes since round 3,
            # SURVEY 8(d)'s algorithmic figure rides along as roofline.frac_algorithmic)
            "hot_path_ms_is": "hot_path.device_ms_graph_replay" if hot_graph_ms is not None else "hot_path.eager_ms_host_bound",
            "hot_path": None if hot_ms is None else {
                "what": "render + warp hot path, forward + backward to the vertices (vertex stage, 2B-mesh flow render, occlusion + "
                        "epilogue, pair loss, their backward passes)",
                "device_ms_graph_replay": None if hot_graph_ms is None else round(hot_graph_ms, 3),
                "eager_ms_host_bound": round(hot_eager_ms, 3)},
            "ranks": ranks, "stock_trunk": stock, "stock_trunk_nchw": stock_nchw, "roofline": roof, "roofline_forward": roof_fwd, "warp_tiles": warp_tiles,
            "in_step_kernels_us": in_step_line, "kernels": kernels, "cpu_baseline": cpu,
        }
        details = args.details_out or "bench_details.json"
        if details != "-":
            try:
                with open(details, "w") as fh:
                    fh.write(json.dumps(line, indent=1) + "\n")
                line["details"] = details
            except OSError as e:  # (a read-only working directory must not cost the run its line)
                sys.stderr.write(f"[bench] could not write {details}: {e}\n")
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(contract_line(line)) + "\n").encode())
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
