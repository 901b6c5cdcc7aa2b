"""TEST / BASELINE INFRASTRUCTURE -- never imported by the product.

One worker PROCESS of bench.py's `cpu_baseline` leg: the CPU oracle (oracle/raster_ref.py = C + OpenMP rasteriser,
oracle/warp_ref.py = numpy) on `--count` images of the hot path -- 2 renders forward, flow masks, occlusion, pair loss
forward + backward, texture backward (kernel E) of both renders; what `/root/reference/meshreg/models/warpbranch.py:9-96`
runs per frame pair, without the encoder.

Why processes and not threads (round 5 ran a thread pool): the numpy half holds the GIL for most of its element-wise
passes and 64 threads x 4 OpenMP threads took 12.8 s per image where one thread takes 1.7 s.  A process per image shares
nothing; the parent (bench.py) starts them, waits until every one has printed "ready" (imports, scene synthesis and the
oracle's .so are behind it), sends "go" to all, and takes the time until the LAST one prints "done": start-up is outside
the timed region, stragglers are inside.

    python -m oracle.cpu_hot_path --seed S --count N --size 256 --omp K
"""
import argparse
import os
import sys
import time


def work(scene, images, is_, omp, R, W, synth, np):
    n = scene["verts1"].shape[0]
    im_ref, im, jm_ref, jm = images
    kw = dict(R=np.eye(3, dtype=np.float32)[None], t=np.zeros((1, 3), np.float32),
              dist_coeffs=np.zeros((1, 5), np.float32), orig_size=is_, image_size=is_, anti_aliasing=False,
              near=0.1, far=100, eps=1e-3, num_threads=omp, keep_saved=True)
    flows, renders = W.get_opticalflow(R, [scene["verts1"], scene["verts2"]], scene["faces"], [scene["K1"], scene["K2"]], kw,
                                       orig_img_size=(is_, is_), ignore_face_idxs=synth.HAND_IGNORE_FACES,
                                       return_renders=True)
    W.pair_consist(flows, im_ref, im, jm_ref, jm, True)
    gl = np.full((n,), 1.0 / 64, np.float32)
    gflows = W.pair_consist_grad(flows, im_ref, im, jm_ref, jm, gl, True)
    # texture backward of the two renders (kernel E; training mode = detach_renders)
    for ro, g in zip(renders, gflows):
        sv = ro["_saved"]
        g_rgb = np.zeros_like(sv["rgb_map"])
        g_rgb[..., :2] = g[:, ::-1]
        R.backward_textures(sv["face_index_map"], sv["sampling_weight_map"], sv["sampling_index_map"], g_rgb,
                            sv["faces"].shape[1], 2)


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--seed", type=int, default=0)
    p.add_argument("--count", type=int, default=1)
    p.add_argument("--size", type=int, default=256)
    p.add_argument("--omp", type=int, default=1)
    a = p.parse_args()
    # the numpy half on this process' one thread (BLAS pools would oversubscribe the box: one process per image already)
    for k in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
        os.environ[k] = str(a.omp)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import numpy as np

    from handobjectconsist_amd.utils import synth  # (numpy only: the seeded scenes bench.py's GPU legs use)
    from oracle import raster_ref as R
    from oracle import warp_ref as W

    scene = synth.random_scene(a.count, seed=a.seed, image_size=a.size)
    images = synth.random_images(a.count, a.size, a.size, a.seed)
    R.lib()
    sys.stdout.write("ready\n")
    sys.stdout.flush()
    if sys.stdin.readline().strip() != "go":
        return 1
    t0 = time.perf_counter()
    work(scene, images, a.size, a.omp, R, W, synth, np)
    sys.stdout.write("done %.4f\n" % (time.perf_counter() - t0))
    sys.stdout.flush()
    return 0


if __name__ == "__main__":
    sys.exit(main())
