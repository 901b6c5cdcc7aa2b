"""``WarpRegNet`` -- counterpart of meshreg/models/warpreg.py:10-127.

Owns the renderer (constructed exactly as warpreg.py:40-51), the closed hand faces, the
photometric criterion and the progressive lambda schedule (warpreg.py:103-110); mixes the
losses as warpreg.py:111-126.  ``step_count`` is deliberately not part of the state dict
(SURVEY Q17)."""
import torch

from handobjectconsist_amd.models import manoutils, warpbranch
from handobjectconsist_amd.neurender import renderer
from handobjectconsist_amd.optim import pyramidloss


def consist_lambdas(step_count, lambda_data, lambda_consist, progressive_consist=True, progressive_steps=1000):
    """(lambda_data_eff, lambda_consist_eff) at a given step (warpreg.py:103-110)."""
    if progressive_consist:
        lc = min(lambda_consist * step_count / progressive_steps, lambda_consist)
        return lambda_data - lc, lc
    return lambda_data, lambda_consist


class WarpRegNet(torch.nn.Module):
    def __init__(
        self,
        image_size,
        model,
        fill_back=True,
        use_backward=True,
        lambda_data=1,
        lambda_consist=1,
        criterion="l1",
        consist_scale=1,
        first_only=True,
        gt_refs=True,
        progressive_consist=True,
        progressive_steps=1000,
        mano_faces=None,
        pair_outputs="full",
    ):
        super().__init__()
        self.fill_back = fill_back
        self.use_backward = use_backward
        max_size = max(image_size)
        self.image_size = image_size
        self.lambda_data = lambda_data
        self.lambda_consist = lambda_consist
        self.consist_scale = consist_scale
        self.criterion = pyramidloss.PyramidCriterion(criterion)
        self.first_only = first_only
        self.progressive_consist = progressive_consist
        self.progressive_steps = progressive_steps
        self.gt_refs = gt_refs
        self.step_count = 0
        self.pair_outputs = pair_outputs
        dev = torch.device("cuda", torch.cuda.current_device())
        self.renderer = renderer.Renderer(
            image_size=max_size,
            R=torch.eye(3, device=dev).unsqueeze(0),
            t=torch.zeros(1, 3, device=dev),
            K=torch.ones(1, 3, 3, device=dev),
            orig_size=max_size,
            anti_aliasing=False,
            fill_back=fill_back,
            near=0.1,
            no_light=True,
            light_intensity_ambient=0.8,
        )
        self.model = model
        if mano_faces is None:
            inner = getattr(model, "module", model)
            mano_faces = inner.mano_layer.th_faces
        closed_faces, hand_ignore_faces = manoutils.get_closed_faces(mano_faces)
        self.hand_ignore_faces = hand_ignore_faces
        self.register_buffer("th_faces", closed_faces, persistent=False)

    def warp_forward(self, samples, all_results):
        return warpbranch.forward(
            samples,
            all_results,
            self.th_faces,
            self.renderer,
            self.image_size,
            self.criterion,
            gt_refs=self.gt_refs,
            first_only=self.first_only,
            hand_ignore_faces=self.hand_ignore_faces,
            use_backward=self.use_backward,
            pair_outputs=self.pair_outputs,
        )

    def forward(self, batch):
        samples = batch["data"]
        all_results, all_losses, mesh_losses = [], [], []
        for sample in samples:
            loss, results, losses = self.model(sample)
            mesh_losses.append(loss)
            all_losses.append(losses)
            all_results.append(results)

        if "consist" in batch["supervision"]:
            warp_loss, pair_results = self.warp_forward(samples, all_results)
        else:
            pair_results = None

        aggregate_losses = {}
        for key in all_losses[0]:
            if all_losses[0][key] is not None:
                aggregate_losses[key] = torch.stack([sample_loss[key] for sample_loss in all_losses]).mean()
        loss = 0
        lambda_data, lambda_consist = consist_lambdas(
            self.step_count, self.lambda_data, self.lambda_consist, self.progressive_consist,
            self.progressive_steps)
        if "data" in batch["supervision"]:
            reg_loss = torch.cat(mesh_losses).mean()
            aggregate_losses["reg_loss"] = reg_loss
            loss += lambda_data * reg_loss
        if "consist" in batch["supervision"]:
            # pose and shape regularization + consistency supervision (warpreg.py:116-126)
            reg_loss = torch.mean(torch.stack([ls["mano_reg_loss"] for ls in all_losses]))
            loss += lambda_data * reg_loss
            loss += lambda_consist * warp_loss
            aggregate_losses["warp_consist"] = warp_loss
            self.step_count += 1
        return loss, aggregate_losses, all_results, pair_results
