"""CPU oracle of the loader-side tensor math  --  TEST INFRASTRUCTURE ONLY.

numpy/torch restatement of reference BH_loader.py:326-329 (buildhir LUT), :361-369 (normalise + clip), :373-392 (label
branch of myImageFloder_S12_globe.__getitem__).  Pinned: ``hierweight`` against the imported reference functions and
the author's comment vectors (tests/golden/g9_hierweight.npz); the per-sample code by executing the reference's own
Dataset class with fake tifffile / cv2 IO (tools/make_golden.py::g_loader, fixture g13_loader.npz: labels, weights and
aggregates bit-exact, normalised tiles to fp32 rounding)."""
import numpy as np
import torch

from oracle.srbh_oracle import aggregate_torch


def buildhir_lut(hir):
    lut = np.zeros((256,), dtype="uint8")                        # :326
    for i in range(len(hir) - 1):
        lut[hir[i]:hir[i + 1]] = i                                # :328-329
    return lut


def label_prep(height_u8, hir, heightweight):
    """one sample (H,W) uint8 -> height, height_aggre, build, weight, weight_aggre (as :373-392)."""
    lut = buildhir_lut(hir)
    heightweight = np.asarray(heightweight)
    build = lut[height_u8]                                        # :374
    weight = heightweight[build]                                  # :375
    height = torch.from_numpy(height_u8).float()                  # :381
    h, w = height.shape
    height_aggre = aggregate_torch(height.reshape((1, 1, h, w)), 0.25)   # :386
    build_aggre = lut[height_aggre.long().numpy()]                # :389
    weight_aggre = heightweight[build_aggre]                      # :390
    return (height, height_aggre, torch.from_numpy(build).long(), torch.from_numpy(weight).float(),
            torch.from_numpy(weight_aggre).float())


def normalize(img_chw, mins, maxs, datarange=(0, 1)):
    img = torch.as_tensor(img_chw).float().clone()
    mins, rng = torch.as_tensor(mins).float(), (torch.as_tensor(maxs) - torch.as_tensor(mins)).float()   # :304-306
    img = (img - mins[:, None, None]) / rng[:, None, None]        # :362-363
    if isinstance(datarange, tuple):
        img[img < datarange[0]] = datarange[0]                    # :368
        img[img > datarange[1]] = datarange[1]                    # :369
    return img
