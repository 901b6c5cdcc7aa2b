"""Batch assembly for meshes of different sizes -- counterpart of ``extend_collate`` /
``seq_extend_collate`` in meshreg/datasets/collate.py:15-36, 77-83 (SURVEY 8f "f4", the part that fixes
the tensor format entering the render path; images / augmentation / dataset indexing stay out of scope).

Objects have different vertex and face counts; the reference pads every per-mesh array of a batch to
the longest one by CYCLIC REPETITION of its rows.  Repeated faces are exact duplicates:
they tie in depth with their originals and the lower face index wins, so the padding never shows in a
render -- the rasteriser's tie rule is what makes this format work.
"""
import numpy as np
import torch


def pad_cyclic(array, length):
    """Rows of `array` repeated cyclically up to `length` rows."""
    array = np.asarray(array)
    if array.shape[0] == 0:
        raise ValueError("cannot pad an empty array")
    reps = -(-length // array.shape[0])
    return np.concatenate([array] * reps)[:length]


def extend_collate(batch, extend_queries=None):
    """List of sample dicts -> dict of batched tensors; the entries named in `extend_queries` are first
    padded to the batch maximum along their first dimension."""
    extend_queries = [q for q in (extend_queries or []) if q in batch[0]]
    sizes = {q: max(np.asarray(sample[q]).shape[0] for sample in batch) for q in extend_queries}
    out = {}
    for key in batch[0]:
        values = [pad_cyclic(sample[key], sizes[key]) if key in sizes else sample[key] for sample in batch]
        first = values[0]
        if torch.is_tensor(first):
            out[key] = torch.stack(values)
        elif isinstance(first, np.ndarray) or isinstance(first, (int, float, np.number)):
            out[key] = torch.from_numpy(np.stack([np.asarray(v) for v in values]))
        else:
            out[key] = values
    return out


def seq_extend_collate(seq, extend_queries=None):
    """Batch of frame sequences (each a list of sample dicts) -> list over frames of collated batches."""
    frames = len(seq[0])
    if any(len(sample_seq) != frames for sample_seq in seq):
        raise ValueError("all sequences of a batch must have the same number of frames")
    return [extend_collate([sample_seq[k] for sample_seq in seq], extend_queries) for k in range(frames)]
