"""PyTorch-ROCm counterpart of the network the hot path trains (bench / trainer only): stock convolutions /
linears (MIOpen, rocBLAS), this build's kernels for what sits between them (frozen BatchNorm + residual + ReLU,
the stem's max-pool, MANO, the head post-processing).

BASELINE.json's north star keeps "the ResNet-18 encoder and regression heads on stock
PyTorch-ROCm"; this module restates their ARCHITECTURE so that the timed optimiser step
does the same work as the reference's (/root/reference/meshreg/models/meshregnet.py:54-384,
resnet.py:92-175, absolutebranch.py, objbranch.py, manobranch.py:11-155, project.py:5-24):
ResNet-18 trunk (features only, the unused `fc` dropped -- SURVEY 8e), two 512x512 MANO
base layers, pose 512->18 (3 global + 15 PCA), shape 512->10, scale/trans branches
512->256->{3,6}, MANO linear-blend skinning, object rotation + weak-perspective recovery,
and the reference's default loss terms (trainmeshwarp.py defaults: recov_joints3d 0.5,
obj recov_verts3d 0.5, pose_reg 5e-6, shape 5e-7).

The MANO model files are licence-gated and absent, so `SynthManoLayer` carries seeded
synthetic parameters of MANO's exact tensor shapes (template [778,3], shapedirs
[778,3,10], posedirs [778,3,135], J_regressor [16,778], weights [778,16], 15 PCA comps)
and performs manopth's `ManoLayer.forward` arithmetic: the batched
contractions run on rocBLAS (MFMA), everything else is element-wise.  Weights are random
(no checkpoints / no network): throughput, not accuracy, is what this model is for.
"""
import os

import numpy as np
import torch
import torch.nn.functional as F
from torch import nn

from handobjectconsist_amd import _lib
from handobjectconsist_amd.nn import frozen_bn
from handobjectconsist_amd.utils import project as camproject
from handobjectconsist_amd.utils import synth


# ----------------------------------------------------------------------------- ResNet-18
# BatchNorm (frozen statistics, --freeze_batchnorm) + residual add + ReLU as one HIP kernel each way instead of
# stock PyTorch's 2-3 element-wise kernels forward and batch_norm_backward + threshold_backward (10.7 of the
# 42 ms of a step).  HOC_HIP_BN=0 / USE_HIP_BN=False: the stock modules.  For fp32 or bf16 (autocast) CUDA
# activations of a module in eval mode; anything else (BatchNorm in training mode, CPU) takes the stock path.
USE_HIP_BN = os.environ.get("HOC_HIP_BN", "1") == "1"


# The trunk's activations and convolution weights in channels-last (NHWC) memory order: MIOpen's fp32 convolutions
# of a step take 23.5 ms there instead of 26.9 ms (no NCHW<->NHWC transposes around its implicit-GEMM kernels,
# scripts/conv_layout.py); the glue kernels above have channels-last variants.  HOC_CHANNELS_LAST=0: NCHW.
USE_CHANNELS_LAST = os.environ.get("HOC_CHANNELS_LAST", "1") == "1"


def _fused_bn(bn, x):
    """x is the convolution's INPUT; the kernels see its output (the BN input), whose dtype is the autocast dtype
    when autocast is on and x's own otherwise.  fp32 and bf16 take the kernels, anything else (fp16 autocast, fp64)
    the stock modules."""
    if not (USE_HIP_BN and not bn.training and x.is_cuda):
        return False
    out_dtype = torch.get_autocast_dtype("cuda") if torch.is_autocast_enabled("cuda") else x.dtype
    return out_dtype in (torch.float32, torch.bfloat16)


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = downsample

    def forward_fused(self, x, x_identity, dup=False):
        """The block on this build's kernels.  x / x_identity: the block input as the two autograd outputs of its
        producer (one for the convolution, one for the identity branch; the same tensor twice is fine too)."""
        residual = x_identity if self.downsample is None else frozen_bn.bn_act(
            self.downsample[0](x_identity), self.downsample[1], relu=False)
        out = frozen_bn.bn_act(self.conv1(x), self.bn1)
        return frozen_bn.bn_act(self.conv2(out), self.bn2, residual=residual, dup=dup)

    def forward(self, x):
        if _fused_bn(self.bn1, x):
            return self.forward_fused(x, x)
        residual = x if self.downsample is None else self.downsample(x)
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.bn2(self.conv2(out))
        return self.relu(out + residual)


class ResNet18Features(nn.Module):
    """resnet.py:92-167 with features=True: conv trunk, global average pool, [B,512]."""

    def __init__(self):
        super().__init__()
        self.inplanes = 64
        self.conv1 = nn.Conv2d(3, 64, kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        self.layer1 = self._make_layer(64, 2)
        self.layer2 = self._make_layer(128, 2, stride=2)
        self.layer3 = self._make_layer(256, 2, stride=2)
        self.layer4 = self._make_layer(512, 2, stride=2)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")

    def _make_layer(self, planes, blocks, stride=1):
        downsample = None
        if stride != 1 or self.inplanes != planes:
            downsample = nn.Sequential(nn.Conv2d(self.inplanes, planes, 1, stride, bias=False),
                                       nn.BatchNorm2d(planes))
        layers = [BasicBlock(self.inplanes, planes, stride, downsample)]
        self.inplanes = planes
        layers += [BasicBlock(planes, planes) for _ in range(1, blocks)]
        return nn.Sequential(*layers)

    def forward(self, x):
        if USE_CHANNELS_LAST and x.is_cuda:
            if not self.conv1.weight.is_contiguous(memory_format=torch.channels_last):
                self.to(memory_format=torch.channels_last)  # once: convolution weights (the Parameter objects stay)
            x = x.contiguous(memory_format=torch.channels_last)
        if _fused_bn(self.bn1, x):
            # every activation with two consumers (the next block's convolution and its identity branch) leaves its
            # producer as two autograd outputs, so that the producer's backward kernel sums the two gradients on load
            blocks = [b for layer in (self.layer1, self.layer2, self.layer3, self.layer4) for b in layer]
            pair = frozen_bn.stem_pool(self.conv1(x), self.bn1, dup=True)  # bn1 + relu + maxpool(3, 2, 1)
            for block in blocks[:-1]:
                pair = block.forward_fused(pair[0], pair[1], dup=True)
            x = blocks[-1].forward_fused(pair[0], pair[1])
        else:
            x = self.maxpool(self.relu(self.bn1(self.conv1(x))))
            x = self.layer4(self.layer3(self.layer2(self.layer1(x))))
        return x.mean(3).mean(2)


class AbsoluteBranch(nn.Module):
    """absolutebranch.py:4-19."""

    def __init__(self, base_neurons=(512, 256), out_dim=3):
        super().__init__()
        layers = []
        for i, o in zip(base_neurons[:-1], base_neurons[1:]):
            layers += [nn.Linear(i, o), nn.ReLU()]
        self.decoder = nn.Sequential(*layers)
        self.final_layer = nn.Linear(base_neurons[-1], out_dim)

    def forward(self, inp):
        return self.final_layer(self.decoder(inp))


# ----------------------------------------------------------------------------- MANO LBS
def batch_rodrigues(rvec):
    """[N,3] axis-angle -> [N,3,3] (manopth rodrigues_layer semantics, via quaternion)."""
    angle = torch.norm(rvec + 1e-8, p=2, dim=1, keepdim=True)
    axis = rvec / angle
    half = angle * 0.5
    q = torch.cat([torch.cos(half), torch.sin(half) * axis], 1)
    q = q / q.norm(p=2, dim=1, keepdim=True)
    w, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    w2, x2, y2, z2 = w * w, x * x, y * y, z * z
    wx, wy, wz, xy, xz, yz = w * x, w * y, w * z, x * y, x * z, y * z
    return torch.stack([w2 + x2 - y2 - z2, 2 * xy - 2 * wz, 2 * wy + 2 * xz, 2 * wz + 2 * xy, w2 - x2 + y2 - z2,
                        2 * yz - 2 * wx, 2 * xz - 2 * wy, 2 * wx + 2 * yz, w2 - x2 - y2 + z2], 1).view(-1, 3, 3)


MANO_PARENTS = [-1, 0, 1, 2, 0, 4, 5, 0, 7, 8, 0, 10, 11, 0, 13, 14]
MANO_TIPS = [745, 317, 444, 556, 673]
MANO_REORDER = [0, 13, 14, 15, 16, 1, 2, 3, 17, 4, 5, 6, 18, 10, 11, 12, 19, 7, 8, 9, 20]


# MANO LBS through the fused HIP kernels (mr_mano_forward / mr_mano_backward: blend-shape GEMM on the matrix
# cores, everything else in three small kernels) instead of ~60 PyTorch ops.  False: the PyTorch restatement
# below (same values to fp32 rounding; it also serves CPU tensors and the non-PCA / tip-centred variants).
USE_HIP_MANO = True


class _ManoLBSFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pose_coeffs, betas, layer):
        ctx.set_materialize_grads(False)
        pose, beta = _lib.contig(pose_coeffs.detach()), _lib.contig(betas.detach())
        B = pose.shape[0]
        c = layer.hip_constants()
        dev = pose.device
        work = torch.empty((max(int(_lib.load().mr_mano_workspace_floats(B)), 1),), dtype=torch.float32, device=dev)
        verts = torch.empty((B, 778, 3), dtype=torch.float32, device=dev)
        jtr = torch.empty((B, 21, 3), dtype=torch.float32, device=dev)
        _lib.call("mr_mano_forward", _lib.ptr(pose), _lib.ptr(beta), *[_lib.ptr(t) for t in c["tensors"]], c["ncomps"],
                  c["center"], _lib.ptr(work), _lib.ptr(verts), _lib.ptr(jtr), B, _lib.stream_ptr(dev))
        ctx.layer, ctx.work, ctx.B = layer, work, B
        return verts, jtr

    @staticmethod
    def backward(ctx, g_verts, g_jtr):
        c = ctx.layer.hip_constants()
        B, dev = ctx.B, ctx.work.device
        g_pose = torch.empty((B, 3 + c["ncomps"]), dtype=torch.float32, device=dev)
        g_beta = torch.empty((B, 10), dtype=torch.float32, device=dev)
        gv = _lib.contig(g_verts) if g_verts is not None else None
        gj = _lib.contig(g_jtr) if g_jtr is not None else None
        _lib.call("mr_mano_backward", *[_lib.ptr(t) for t in c["tensors"]], c["ncomps"], c["center"], _lib.ptr(ctx.work),
                  _lib.ptr(gv), _lib.ptr(gj), _lib.ptr(g_pose), _lib.ptr(g_beta), B, _lib.stream_ptr(dev))
        return g_pose, g_beta, None


class SynthManoLayer(nn.Module):
    """manopth ManoLayer.forward on synthetic MANO-shaped parameters."""

    def __init__(self, ncomps=15, use_pca=True, flat_hand_mean=False, center_idx=9, seed=0, torch_variants=False):
        super().__init__()
        # GPU calls the HIP kernels do not cover (axis-angle input, a translation, a finger-tip centre, non-fp32) RAISE
        # unless the layer was built with torch_variants=True: nothing on a GPU falls back to PyTorch silently
        self.torch_variants = torch_variants
        rng = np.random.default_rng(seed)
        v, f = synth.hand_template()
        self.use_pca, self.ncomps, self.center_idx = use_pca, ncomps, center_idx
        # 16 synthetic joints spread through the template, soft skinning weights by distance
        jpos = v[rng.choice(v.shape[0], 16, replace=False)] * 0.6
        d = ((v[:, None] - jpos[None]) ** 2).sum(-1)
        wts = np.exp(-d / (2 * 0.02 ** 2))
        wts = wts / wts.sum(1, keepdims=True)
        jreg = np.exp(-d.T / (2 * 0.01 ** 2))
        jreg = jreg / jreg.sum(1, keepdims=True)
        comps = np.linalg.qr(rng.standard_normal((45, 45)))[0]
        buf = lambda name, a: self.register_buffer(name, torch.tensor(np.asarray(a), dtype=torch.float32))
        buf("th_v_template", v[None])
        buf("th_shapedirs", rng.standard_normal((778, 3, 10)) * 0.002)
        buf("th_posedirs", rng.standard_normal((778, 3, 135)) * 0.0005)
        buf("th_J_regressor", jreg)
        buf("th_weights", wts)
        buf("th_comps", comps)
        buf("th_hands_mean", np.zeros((1, 45)) if flat_hand_mean else rng.standard_normal((1, 45)) * 0.1)
        self.register_buffer("th_faces", torch.tensor(f[:1538], dtype=torch.long))
        # constants of forward() as buffers: nothing is built from host data per call (hipGraph-capturable)
        self.register_buffer("_parents", torch.tensor(MANO_PARENTS[1:], dtype=torch.long), persistent=False)
        self.register_buffer("_eye3", torch.eye(3), persistent=False)
        self.register_buffer("_row0001", torch.tensor([0.0, 0.0, 0.0, 1.0]), persistent=False)
        self.register_buffer("_tips", torch.tensor(MANO_TIPS, dtype=torch.long), persistent=False)
        self.register_buffer("_reorder", torch.tensor(MANO_REORDER, dtype=torch.long), persistent=False)
        for name, idx in (("_lvl1", [0, 3, 6, 9, 12]), ("_lvl2", [1, 4, 7, 10, 13]), ("_lvl3", [2, 5, 8, 11, 14])):
            self.register_buffer(name, torch.tensor(idx, dtype=torch.long), persistent=False)  # indices into rel (joint - 1)

    def hip_constants(self):
        """Model constants in the layout of mr_mano_forward, built once per device."""
        dev = self.th_v_template.device
        sources = (self.th_v_template, self.th_shapedirs, self.th_posedirs, self.th_J_regressor, self.th_weights,
                   self.th_comps, self.th_hands_mean)
        key = (dev, self.ncomps, self.center_idx) + tuple((b.data_ptr(), b._version) for b in sources)
        cached = getattr(self, "_hip_consts", None)
        if cached is not None and cached["key"] == key:  # rebuilt if the buffers move or are overwritten
            return cached
        with torch.no_grad():
            blend = torch.zeros((146, 2334), dtype=torch.float32, device=dev)
            blend[:10] = self.th_shapedirs.reshape(2334, 10).t()
            blend[10:145] = self.th_posedirs.reshape(2334, 135).t()
            js = torch.matmul(self.th_J_regressor, self.th_shapedirs.reshape(778, 30)).view(48, 10).contiguous()
            jt = torch.matmul(self.th_J_regressor, self.th_v_template[0]).reshape(48).contiguous()
            i32 = lambda values: torch.tensor(values, dtype=torch.int32, device=dev)
            center = -1 if self.center_idx is None else MANO_REORDER[self.center_idx]
            tensors = [self.th_comps[: self.ncomps].contiguous(), self.th_hands_mean.reshape(45).contiguous(), js, jt,
                       blend, self.th_v_template.reshape(2334).contiguous(), self.th_weights.contiguous(),
                       i32(MANO_PARENTS), i32(MANO_TIPS), i32(MANO_REORDER)]
        self._hip_consts = {"key": key, "tensors": tensors, "ncomps": self.ncomps, "center": center}
        return self._hip_consts

    def _hip_path(self, th_pose_coeffs, th_betas, th_trans):
        center_ok = self.center_idx is None or MANO_REORDER[self.center_idx] < 16
        return (USE_HIP_MANO and th_pose_coeffs.is_cuda and th_pose_coeffs.dtype == torch.float32 and self.use_pca
                and th_trans is None and center_ok and th_pose_coeffs.shape[1] == 3 + self.ncomps)

    def forward(self, th_pose_coeffs, th_betas=None, th_trans=None):
        """GPU: the HIP kernels (mr_mano_forward / _backward).  CPU tensors (no device: the gloo tests, host-side
        tools) take the PyTorch restatement below.  A GPU call outside the kernels' coverage raises unless the layer
        was built with ``torch_variants=True`` (or USE_HIP_MANO was switched off for an A/B): no silent fallback."""
        if self._hip_path(th_pose_coeffs, th_betas, th_trans):
            if th_betas is None:
                th_betas = th_pose_coeffs.new_zeros((th_pose_coeffs.shape[0], 10))
            return _ManoLBSFunction.apply(th_pose_coeffs, th_betas, self)
        if th_pose_coeffs.is_cuda and USE_HIP_MANO and not self.torch_variants:
            raise RuntimeError(
                "SynthManoLayer: no HIP kernel for this call (needs fp32 PCA coefficients [B, 3 + ncomps], no translation, a "
                "centre among the 16 articulated joints); build the layer with torch_variants=True to run the PyTorch "
                "restatement on the GPU instead")
        return self.forward_torch(th_pose_coeffs, th_betas, th_trans)

    def forward_torch(self, th_pose_coeffs, th_betas=None, th_trans=None):
        """PyTorch restatement (CPU tensors; GPU only on request, see ``forward``), checked against oracle/mano_ref.py.
        Same contractions as manopth, arranged as a few dense GEMMs (rocBLAS /
        MFMA) instead of many tiny batched ones: blend shapes as ONE [B,145] x [145,2334] product,
        joints from pre-multiplied regressors, the kinematic chain level by level (3 batched
        products instead of 15 sequential ones), skinning as one [778,16] x [16,B*16] product
        and the final per-vertex 4x4 transform as a broadcast multiply-sum."""
        B = th_pose_coeffs.shape[0]
        hand = th_pose_coeffs[:, 3:3 + self.ncomps] if self.use_pca else th_pose_coeffs[:, 3:]
        full_hand = hand.mm(self.th_comps[: self.ncomps]) if self.use_pca else hand
        full_pose = torch.cat([th_pose_coeffs[:, :3], self.th_hands_mean + full_hand], 1)
        rots = batch_rodrigues(full_pose.reshape(-1, 3)).view(B, 16, 3, 3)
        pose_map = (rots[:, 1:] - self._eye3).reshape(B, 135)
        if th_betas is None:
            th_betas = th_pose_coeffs.new_zeros((B, 10))
        blend = torch.cat([self.th_shapedirs.reshape(2334, 10), self.th_posedirs.reshape(2334, 135)], 1)  # [2334,145]
        v_posed = (torch.cat([th_betas, pose_map], 1) @ blend.t()).view(B, 778, 3) + self.th_v_template
        js = torch.matmul(self.th_J_regressor, self.th_shapedirs.reshape(778, 30)).view(48, 10)  # J_reg @ shapedirs
        jt = torch.matmul(self.th_J_regressor, self.th_v_template[0])                             # [16,3]
        joints = (th_betas @ js.t()).view(B, 16, 3) + jt

        def with_zeros(rot, tr):  # [..,3,3], [..,3] -> [..,4,4]
            top = torch.cat([rot, tr.unsqueeze(-1)], -1)
            bottom = self._row0001.expand(*top.shape[:-2], 1, 4)
            return torch.cat([top, bottom], -2)

        parents = self._parents
        rel = with_zeros(rots[:, 1:], joints[:, 1:] - joints.index_select(1, parents))          # [B,15,4,4]
        root = with_zeros(rots[:, 0], joints[:, 0])                                # [B,4,4]
        g1 = torch.matmul(root.unsqueeze(1), rel.index_select(1, self._lvl1))
        g2 = torch.matmul(g1, rel.index_select(1, self._lvl2))
        g3 = torch.matmul(g2, rel.index_select(1, self._lvl3))
        G = torch.cat([root.unsqueeze(1), torch.stack([g1, g2, g3], 2).reshape(B, 15, 4, 4)], 1)  # joints 1..15 in order
        j_h = torch.cat([joints, joints.new_zeros((B, 16, 1))], 2).unsqueeze(-1)
        G2 = G - F.pad(torch.matmul(G, j_h), (3, 0))
        T = (self.th_weights @ G2.permute(1, 0, 2, 3).reshape(16, B * 16)).view(778, B, 4, 4).permute(1, 0, 2, 3)
        v_h = torch.cat([v_posed, v_posed.new_ones((B, 778, 1))], 2)
        verts = (T[:, :, :3, :] * v_h.unsqueeze(2)).sum(-1)
        jtr = torch.cat([G[:, :, :3, 3], verts.index_select(1, self._tips)], 1).index_select(1, self._reorder)
        if th_trans is None:
            if self.center_idx is not None:
                center = jtr[:, self.center_idx].unsqueeze(1)
                jtr, verts = jtr - center, verts - center
        else:
            jtr, verts = jtr + th_trans.unsqueeze(1), verts + th_trans.unsqueeze(1)
        return verts * 1000, jtr * 1000


# ----------------------------------------------------------------------------- the network
def recover_3d_proj(objpoints3d, camintr, est_scale, est_trans, off_z=0.4, input_res=(128, 128)):
    """meshreg/models/project.py:5-24 (pinned by tests/golden/warp_misc.npz)."""
    focal = camintr[:, :1, :1]
    batch_size = objpoints3d.shape[0]
    focal = focal.view(batch_size, 1)
    est_scale = est_scale.view(batch_size, 1)
    est_trans = est_trans.view(batch_size, 2)
    est_Z0 = focal * est_scale + off_z
    cam_centers = camintr[:, :2, 2]
    # (input_res / 2 as two scalar fills: no host array -> device copy per call)
    img_centers = torch.stack([cam_centers.new_full((batch_size,), input_res[0] / 2),
                               cam_centers.new_full((batch_size,), input_res[1] / 2)], 1)
    est_XY0 = (est_trans + img_centers - cam_centers) * est_Z0 / focal
    est_c3d = torch.cat([est_XY0, est_Z0], -1).unsqueeze(1)
    return est_c3d + objpoints3d, est_c3d


# recover_3d_proj x2, unit conversion, object rotation and both projections through ONE HIP kernel each way
# (mr_meshreg_post_forward / _backward) instead of ~280 small PyTorch launches per step.  False: op by op.
USE_HIP_POST = True


class _MeshRegPostFunction(torch.autograd.Function):
    """(verts_mm, joints_mm, scaletrans, st_obj; camintr, objcanverts) -> (recov_handverts3d, recov_joints3d,
    joints2d, recov_objverts3d, obj_verts2d)."""

    @staticmethod
    def forward(ctx, verts_mm, joints_mm, scaletrans, st_obj, camintr, canverts, trans_factor, scale_factor, input_res):
        ctx.set_materialize_grads(False)
        c = [_lib.contig(x.detach()) for x in (verts_mm, joints_mm, scaletrans, st_obj, camintr, canverts)]
        B, Vh, J, Vo = c[0].shape[0], c[0].shape[1], c[1].shape[1], c[5].shape[1]
        dev = c[0].device
        new = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)
        outs = [new(B, Vh, 3), new(B, J, 3), new(B, J, 2), new(B, Vo, 3), new(B, Vo, 2)]
        ctx.consts = (float(trans_factor), float(scale_factor), 0.4, float(input_res[0]), float(input_res[1]))
        _lib.call("mr_meshreg_post_forward", *[_lib.ptr(x) for x in c], *ctx.consts, *[_lib.ptr(o) for o in outs],
                  B, Vh, J, Vo, _lib.stream_ptr(dev))
        ctx.save_for_backward(*c)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *grads):
        c = ctx.saved_tensors
        B, Vh, J, Vo = c[0].shape[0], c[0].shape[1], c[1].shape[1], c[5].shape[1]
        dev = c[0].device
        g = [_lib.contig(x) if x is not None else None for x in grads]
        work = torch.empty((B * 15,), dtype=torch.float32, device=dev)
        outs = [torch.empty_like(c[0]), torch.empty_like(c[1]), torch.empty_like(c[2]), torch.empty_like(c[3])]
        _lib.call("mr_meshreg_post_backward", *[_lib.ptr(x) for x in c], *ctx.consts, *[_lib.ptr(x) for x in g],
                  _lib.ptr(work), *[_lib.ptr(o) for o in outs], B, Vh, J, Vo, _lib.stream_ptr(dev))
        return outs[0], outs[1], outs[2], outs[3], None, None, None, None, None


def cat_or_view(tensors):
    """``torch.cat(tensors)`` along dim 0 -- without the copy when the tensors already ARE consecutive slices of one
    contiguous buffer (the frames of a step as `mr_frames_to_batch` or the synthetic loader lay them out: the
    150 MB image concatenation of a B = 64 step then costs nothing)."""
    first = tensors[0]
    if all(t.is_contiguous() and t.dtype == first.dtype and t.device == first.device and t.shape[1:] == first.shape[1:]
           for t in tensors):
        base = first.untyped_storage().data_ptr()
        end, ok = first.data_ptr(), True
        for t in tensors:
            ok = ok and t.untyped_storage().data_ptr() == base and t.data_ptr() == end
            end += t.numel() * t.element_size()
        if ok:
            rows = sum(t.shape[0] for t in tensors)
            return first.as_strided((rows,) + tuple(first.shape[1:]), first.stride(), first.storage_offset())
    return torch.cat(tensors)


class SynthMeshRegNet(nn.Module):
    """MeshRegNet (meshregnet.py:54-384) with the trainmeshwarp.py default loss weights.

    forward(sample) -> (total_loss [1], results, losses); sample is a dict of device tensors:
    image [B,3,H,W], camintr [B,3,3], objcanverts [B,Vo,3], and (supervised frames only)
    joints3d [B,21,3], objverts3d [B,Vo,3]."""

    def __init__(self, obj_trans_factor=100, obj_scale_factor=0.0001, lambda_recov_joints3d=0.5,
                 lambda_obj_recov_verts3d=0.5, lambda_pose_reg=5e-6, lambda_shape=5e-7, mano_comps=15):
        super().__init__()
        self.base_net = ResNet18Features()
        self.scaletrans_branch = AbsoluteBranch((512, 256), 3)
        self.scaletrans_branch_obj = AbsoluteBranch((512, 256), 6)
        self.mano_base = nn.Sequential(nn.Linear(512, 512), nn.ReLU(), nn.Linear(512, 512), nn.ReLU())
        self.pose_reg = nn.Linear(512, mano_comps + 3)
        self.shape_reg = nn.Linear(512, 10)
        self.mano_layer = SynthManoLayer(ncomps=mano_comps, use_pca=True, center_idx=9)
        self.obj_trans_factor, self.obj_scale_factor = obj_trans_factor, obj_scale_factor
        self.lam = (lambda_recov_joints3d, lambda_obj_recov_verts3d, lambda_pose_reg, lambda_shape)
        # BASELINE.json config 5 ("bf16"): the TRUNK under bf16 autocast; heads, MANO, render and warp stay fp32
        self.encoder_dtype = torch.float32
        if USE_CHANNELS_LAST:
            # convolution weights in channels-last from the start, i.e. before a DistributedDataParallel wrapper
            # builds its gradient buckets from the parameters' strides
            self.base_net.to(memory_format=torch.channels_last)

    def encode(self, images):
        """ResNet-18 trunk -> [B,512] fp32 features (optionally computed under bf16 autocast)."""
        if self.encoder_dtype == torch.float32:
            return self.base_net(images)
        with torch.autocast("cuda", dtype=self.encoder_dtype):
            feats = self.base_net(images)
        return feats.float()

    def encode_frames(self, samples):
        """ONE encoder pass over the frames of several samples (same resolution): the reference runs
        the ResNet once per frame (warpreg.py:86-90); with the BatchNorm statistics frozen
        (--freeze_batchnorm) every layer acts per image, so the concatenated pass gives the same
        features with a third of the launches.  The features are left in
        ``sample["_features"]`` for the following ``forward(sample)`` calls."""
        if self.base_net.training:
            raise RuntimeError("encode_frames needs frozen BatchNorm statistics (model.eval())")
        sizes = [s["image"].shape[0] for s in samples]
        feats = self.encode(cat_or_view([s["image"] for s in samples]))
        for s, f in zip(samples, feats.split(sizes)):
            s["_features"] = f

    def heads(self, features):
        """The four regression heads: the only trainable part after the trunk."""
        base = self.mano_base(features)
        return self.pose_reg(base), self.shape_reg(base), self.scaletrans_branch(features), self.scaletrans_branch_obj(features)

    def post_heads(self, pose, shape, scaletrans, st_obj, camintr, objcanverts, input_res=(256, 256)):
        """Head outputs -> hand / object meshes in the camera frame and their projections.  No trainable
        parameter is read here (MANO and the camera recovery are fixed functions), every operation acts
        per sample, and nothing is built from host data per call: a host -> device copy of a constant is
        a hidden synchronisation (removing them was worth 10 % of the step).  Replaying this function as a
        hipGraph per frame (torch.cuda.make_graphed_callables) was tried and measured SLOWER than the
        eager launches (52.2 vs 47.3 ms per step); what does pay is running it ONCE over all frames of
        a step (``prepare_frames``)."""
        # hand: MANO branch (manobranch.py:88-155) + camera recovery (meshregnet.py:206-245)
        verts, joints = self.mano_layer(pose, th_betas=shape)
        if (USE_HIP_POST and verts.is_cuda and verts.dtype == torch.float32 and camintr.shape[0] == verts.shape[0]
                and objcanverts.shape[0] == verts.shape[0]):
            return _MeshRegPostFunction.apply(verts, joints, scaletrans, st_obj, camintr, objcanverts,
                                              self.obj_trans_factor, self.obj_scale_factor, input_res)
        verts3d, joints3d = verts / 1000, joints / 1000
        trans, scale = scaletrans[:, 1:], scaletrans[:, :1]
        final_trans = trans.unsqueeze(1) * self.obj_trans_factor
        final_scale = scale.view(-1, 1, 1) * self.obj_scale_factor
        recov_joints3d, center3d = recover_3d_proj(joints3d, camintr, final_scale, final_trans, input_res=input_res)
        recov_handverts3d = verts3d + center3d
        joints2d = camproject.batch_proj2d(recov_joints3d, camintr)
        # object: rotation + weak-perspective recovery (objbranch.py:28-84, meshregnet.py:274-323)
        rotmat = batch_rodrigues(st_obj[:, 3:])
        rotobjverts = rotmat.bmm(objcanverts.transpose(1, 2)).transpose(1, 2)
        o_trans = st_obj[:, 1:3].unsqueeze(1) * self.obj_trans_factor
        o_scale = st_obj[:, :1].view(-1, 1, 1) * self.obj_scale_factor
        objverts3d, _ = recover_3d_proj(rotobjverts, camintr, o_scale, o_trans, input_res=input_res)
        obj_verts2d = camproject.batch_proj2d(objverts3d, camintr)
        return recov_handverts3d, recov_joints3d, joints2d, objverts3d, obj_verts2d

    def prepare_frames(self, samples, batch_encoder=False):
        """Everything of ``forward`` that does not depend on the supervision, for ALL frames of an
        optimiser step at once: the encoder (one pass per frame as in the reference, or one pass over
        the concatenation), then ONE pass of the heads and of ``post_heads`` over the concatenated
        features -- a third of their ~250 launches per frame (forward + backward).  Per-sample
        operations only, so each frame's slice equals what its own ``forward`` would compute.  Returns
        one tuple per frame, to be passed back as ``sample["_post"]``; the loss terms stay per frame in
        ``forward``."""
        if batch_encoder:
            if self.base_net.training:
                raise RuntimeError("a single encoder pass needs frozen BatchNorm statistics (model.eval())")
            sizes_img = [s["image"].shape[0] for s in samples]
            feats = list(self.encode(cat_or_view([s["image"] for s in samples])).split(sizes_img))
        else:
            feats = [self.encode(s["image"]) for s in samples]
        sizes = [f.shape[0] for f in feats]
        H, W = samples[0]["image"].shape[2:]
        pose, shape, scaletrans, st_obj = self.heads(torch.cat(feats))
        geo = self.post_heads(pose, shape, scaletrans, st_obj, torch.cat([s["camintr"] for s in samples]),
                              torch.cat([s["objcanverts"] for s in samples]), input_res=(W, H))
        # the pose / shape regularisers of all frames in three launches instead of ~12 per frame: per-frame means
        # (mse against zero = mean of squares), same values as the per-frame expressions in `forward`
        if len(set(sizes)) == 1:
            n, b = len(sizes), sizes[0]
            regs = (self.lam[3] * shape.reshape(n, b, -1).square().mean((1, 2))
                    + self.lam[2] * pose[:, 3:].reshape(n, b, -1).square().mean((1, 2))).unbind(0)
        else:
            regs = [None] * len(sizes)
        # returned, not stored: behind a DistributedDataParallel wrapper `samples` may be re-built copies of
        # the caller's containers (DDP moves inputs to its device recursively), so the caller stashes them
        return [frame + (reg,) for frame, reg in zip(zip(*[t.split(sizes) for t in geo + (pose, shape)]), regs)]

    def forward(self, sample, no_loss=False, encode_only=False, batch_encoder=False):
        if encode_only:  # (through forward so that a DistributedDataParallel wrapper sees the call)
            return self.prepare_frames(sample, batch_encoder=batch_encoder)
        image = sample["image"]
        H, W = image.shape[2:]
        supervised = not no_loss and "joints3d" in sample and "objverts3d" in sample
        if not no_loss and not supervised and ("joints3d" in sample or "objverts3d" in sample):
            raise ValueError("a supervised frame carries both joints3d and objverts3d")
        post = sample.get("_post")
        if post is None:
            features = sample.get("_features")
            if features is None:
                features = self.encode(image)
            pose, shape, scaletrans, st_obj = self.heads(features)
            post = self.post_heads(pose, shape, scaletrans, st_obj, sample["camintr"], sample["objcanverts"],
                                   input_res=(W, H)) + (pose, shape)
        results = dict(zip(("recov_handverts3d", "recov_joints3d", "joints2d", "recov_objverts3d", "obj_verts2d"), post[:5]))
        pose, shape = post[5], post[6]
        lam_j, lam_o, lam_pose, lam_shape = self.lam
        if len(post) > 7 and post[7] is not None:
            reg_loss = post[7]  # computed for all frames at once by prepare_frames
        else:
            reg_loss = lam_shape * F.mse_loss(shape, torch.zeros_like(shape)) \
                + lam_pose * F.mse_loss(pose[:, 3:], torch.zeros_like(pose[:, 3:]))
        losses = {"mano_reg_loss": reg_loss.view(1)}
        total_loss = reg_loss.view(1)
        if supervised:
            losses["recov_joint3d"] = F.mse_loss(results["recov_joints3d"], sample["joints3d"])
            total_loss = total_loss + lam_j * losses["recov_joint3d"]
            losses["recov_objverts3d"] = F.mse_loss(results["recov_objverts3d"], sample["objverts3d"])
            total_loss = total_loss + lam_o * losses["recov_objverts3d"]
        losses["total_loss"] = total_loss
        return total_loss, results, losses
