#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/c7
mkdir -p $OUT
cd $ROOT
export HOC_KERNEL_GROUPS="render_backward_full(D+E+F);render_backward_train(E)"
for f in 0 64; do
  export HOC_BWD_FLAGS=$((f << 8))
  bash scripts/prof_kernels.sh c7f$f $ROOT/bench.py --kernels-only > /dev/null 2>&1
  echo "== flags>>8 = $f"; grep -E "gather_kernel|pixel_map_strip|compact" $ROOT/gpurun_out/prof_c7f${f}_by_grid.txt
done
python - <<'PY'
# how many faces of the bench scene take the gather's whole-wave path (bbox above 256 pixels), and how large are they
import sys, numpy as np
sys.path.insert(0, ".")
from handobjectconsist_amd.utils import synth
s = synth.random_scene(64, seed=0, image_size=256)
K, v, f = s["K1"], s["verts1"], s["faces"]
p = np.einsum("bij,bvj->bvi", K, v); p = p[..., :2] / p[..., 2:3]
tri = np.take_along_axis(p[:, :, None, :].repeat(1, 2), f[..., None].repeat(2, -1).reshape(64, -1, 3, 2)[:, :, :, :1].astype(np.int64) * 0 + 0, 1) if False else None
pf = np.stack([np.stack([p[b][f[b][:, k]] for k in range(3)], 1) for b in range(64)])  # [B,F,3,2]
w = pf[..., 0].max(-1) - pf[..., 0].min(-1); h = pf[..., 1].max(-1) - pf[..., 1].min(-1)
area = (w + 1) * (h + 1)
print("faces", area.shape, "bbox area > 256:", int((area > 256).sum()), "per image", (area > 256).sum(1)[:8], "max area", float(area.max()))
big = np.argwhere(area > 256)
print("face indices of image 0 with big boxes:", big[big[:, 0] == 0][:, 1][:40])
PY
