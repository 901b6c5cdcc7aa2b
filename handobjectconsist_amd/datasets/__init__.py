"""Dataset -> batch pipeline either side of the render + warp path: spatial augmentation
parameters and 2-D / intrinsics transforms on the host (``handutils``), the affine crop + tensorisation +
jitter mask of a whole batch of decoded frames on the GPU (``frames.frames_to_batch``), and the
sequence-sampling dataset wrapper (``handobjset.HandObjSet``)."""
