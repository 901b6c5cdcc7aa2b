"""Photometric-consistency branch -- counterpart of meshreg/models/warpbranch.py:9-96.

Same argument list and control flow as the reference's ``forward``: GT-reference
substitution (:38-47), hand + object mesh concatenation (:49-55), detach of frames > 0,
``get_opticalflows(..., detach_textures=False, detach_renders=True)`` (:59-68) and one
``pair_consist`` per frame pair (:73-88).  Samples are dicts of device-resident tensors
keyed by plain strings (the dataset layer and its Queries enums are out of scope)."""
import torch

from handobjectconsist_amd.utils import catmesh
from handobjectconsist_amd.warping import imgflowarp, opticalflow


def forward(
    samples,
    all_results,
    hand_face,
    renderer,
    image_size,
    criterion,
    gt_refs=True,
    first_only=True,
    hand_ignore_faces=None,
    use_backward=True,
    pair_outputs="full",
):
    """
    Args:
        use_backward: also compare the warp from the unannotated to the annotated frame with
            the annotated image
        pair_outputs: "full" (masks / warps / diffs as the reference) or "loss"
    """
    images = [sample["image"].cuda() for sample in samples]
    jitter_masks = [sample["jittermask"].cuda() for sample in samples]
    camintrs = [sample["camintr"].cuda() for sample in samples]

    obj_verts = [result["recov_objverts3d"] for result in all_results]
    obj_faces = [sample["objfaces"].long().cuda() for sample in samples]
    hand_verts = [result["recov_handverts3d"] for result in all_results]
    hand_faces_b = hand_face.repeat(obj_verts[0].shape[0], 1, 1).long()
    hand_faces = [hand_faces_b for _ in range(len(samples))]
    if gt_refs:
        # Replace reference vertices by ground truth vertices (warpbranch.py:38-47)
        for sample_idx in range(1, len(samples)):
            obj_verts[sample_idx] = samples[sample_idx]["objverts3d"].cuda()
            hand_verts[sample_idx] = samples[sample_idx]["handverts3d"].cuda()
    verts_world = []
    for seq_idx in range(len(samples)):
        all_verts, all_faces, _ = catmesh.batch_cat_meshes(
            [hand_verts[seq_idx], obj_verts[seq_idx]], [hand_faces[seq_idx], obj_faces[seq_idx]]
        )
        if first_only and seq_idx > 0:
            all_verts = all_verts.detach()
        verts_world.append(all_verts)

    recons_flows = opticalflow.get_opticalflows(
        verts_world,
        all_faces,
        camintrs,
        renderer,
        image_size,
        detach_textures=False,
        detach_renders=True,
        ignore_face_idxs=hand_ignore_faces,
    )
    all_masks, all_warps, all_diffs, full_losses = [], [], [], []
    for recons_flow, image, jitter_mask in zip(recons_flows, images[1:], jitter_masks[1:]):
        warp_loss, masks, warps, diffs = imgflowarp.pair_consist(
            recons_flow,
            image_ref=images[0],
            image=image,
            jitter_mask_ref=jitter_masks[0],
            jitter_mask=jitter_mask,
            criterion=criterion,
            use_backward=use_backward,
            outputs=pair_outputs,
        )
        all_masks.append(masks)
        full_losses.append(warp_loss)
        all_warps.append(warps)
        all_diffs.append(diffs)
    stack_losses = torch.stack(full_losses)
    full_loss = stack_losses.mean()
    pair_results = {
        "masks": all_masks,
        "warps": all_warps,
        "recons_flows": recons_flows,
        "diffs": all_diffs,
        "diff_losses": stack_losses,
    }
    return full_loss, pair_results
