"""Keys of the per-frame sample dicts -- counterpart of meshreg/datasets/queries.py:4-46 (same enum and
member names in the same order, so ``auto()`` numbering agrees).  ``BaseQueries`` name what a dataset
holds in its own frame, ``TransQueries`` the same quantities after the training-time augmentation."""
from enum import Enum

BaseQueries = Enum("BaseQueries", [
    "CAMINTR", "OBJFACES", "OBJCORNERS2D", "OBJCORNERS3D", "OBJVERTS3D", "OBJVERTS2D", "OBJVIS2D", "HANDVERTS3D",
    "HANDVERTS2D", "HANDVIS2D", "JOINTS3D", "JOINTS2D", "IMAGE", "SIDE", "OBJCANVERTS", "OBJCANROTVERTS",
    "OBJCANROTCORNERS", "OBJCANSCALE", "OBJCANTRANS", "OBJCANCORNERS", "JOINTVIS"], module=__name__)

TransQueries = Enum("TransQueries", [
    "CAMINTR", "OBJVERTS3D", "OBJVERTS2D", "OBJCORNERS2D", "OBJCORNERS3D", "OBJCANROTVERTS", "OBJCANROTCORNERS",
    "HANDVERTS3D", "HANDVERTS2D", "JOINTS3D", "JOINTS2D", "CENTER3D", "IMAGE", "JITTERMASK", "SIDE", "SCALE",
    "AFFINETRANS", "ROTMAT"], module=__name__)

# what warpbranch.forward reads from a sample (warpbranch.py:28-47) and the plain-string spelling the
# synthetic loaders of this package use for the same entries
PATH_KEYS = {
    "image": TransQueries.IMAGE,
    "jittermask": TransQueries.JITTERMASK,
    "camintr": TransQueries.CAMINTR,
    "objfaces": BaseQueries.OBJFACES,
    "objverts3d": BaseQueries.OBJVERTS3D,
    "handverts3d": BaseQueries.HANDVERTS3D,
}


def lookup(sample, name):
    """``sample[...]`` by plain string, by this module's enum member, or by the same-named member of a
    same-named enum class from another module (the reference's own queries module)."""
    if name in sample:
        return sample[name]
    query = PATH_KEYS[name]
    if query in sample:
        return sample[query]
    for key in sample:
        if getattr(key, "name", None) == query.name and type(key).__name__ == type(query).__name__:
            return sample[key]
    raise KeyError(f"sample has no entry for {query} / '{name}'")
