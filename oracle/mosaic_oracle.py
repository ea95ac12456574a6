"""CPU oracle of the inference epilogue  --  TEST INFRASTRUCTURE ONLY.

Literal numpy restatement of reference predict_realesanet_feature_globe.py:156-158,172-185,195-204 (same dtypes, same
operation order).  Pinned: tools/make_golden.py executes the reference's own ``predict_whole_image_grid`` (module
imported with stub GIS / IO modules, grid loader + raster writers + the two networks replaced by fakes that feed the
synthetic city below) and stores what it hands to the raster writers (tests/golden/g12_mosaic.npz); this oracle and the
HIP kernels must reproduce those arrays bit for bit."""
import numpy as np
import torch


class MosaicOracle:
    def __init__(self, height, width, chans_build):
        self.res_height = np.zeros((height, width), dtype=np.uint16)               # :156
        self.res_build = np.zeros((chans_build, height, width), dtype=np.uint16)   # :157
        self.res_weight = np.zeros((height, width), dtype=np.uint8)                # :158

    def add(self, ypred, build_pred, posall):
        ypred = ypred.cpu().numpy().copy()                                          # :172
        ypred[ypred < 0] = 0                                                        # :173
        ypred = np.round(ypred * 10).astype(np.uint16)                              # :174
        build = torch.softmax(build_pred, dim=1).cpu().numpy()                      # :176
        build = np.round(build * 255).astype(np.uint16)                             # :177
        for i in range(ypred.shape[0]):                                             # :181-185
            xoff, yoff, xcount, ycount = (np.asarray(posall[i]) * 4).tolist()
            self.res_height[yoff:yoff + ycount, xoff:xoff + xcount] += ypred[i, 0, :ycount, :xcount]
            self.res_build[:, yoff:yoff + ycount, xoff:xoff + xcount] += build[i, :, :ycount, :xcount]
            self.res_weight[yoff:yoff + ycount, xoff:xoff + xcount] += 1

    def finalize(self):
        build = np.argmax(self.res_build, axis=0).astype(np.uint8)                  # :195
        h = self.res_height.copy()
        mask = self.res_weight > 0                                                  # :201
        h[mask] = np.round(h[mask] / self.res_weight[mask]).astype(np.uint16)       # :203
        return h, build


def synthetic_city(seed=2024, lr_w=24, lr_h=20, tile=8, n_extra=14, chans_build=7):
    """A small 'city' for the mosaic fixtures: LR raster lr_h x lr_w, tiles of `tile` LR cells (4x that in HR), a regular
    grid of windows plus `n_extra` random overlapping ones; windows at the right/bottom edge are clipped (xcount/ycount <
    tile) as the reference's grid index does.  Returns per-tile predictions (N,1,4t,4t) / logits (N,C,4t,4t) and the
    (N,4) int64 LR windows [xoff, yoff, xcount, ycount]."""
    g = torch.Generator()
    g.manual_seed(seed)
    pos = []
    for y in range(0, lr_h, tile):
        for x in range(0, lr_w, tile):
            pos.append([x, y, min(tile, lr_w - x), min(tile, lr_h - y)])
    for _ in range(n_extra):
        x = int(torch.randint(0, lr_w - 1, (1,), generator=g))
        y = int(torch.randint(0, lr_h - 1, (1,), generator=g))
        pos.append([x, y, min(tile, lr_w - x), min(tile, lr_h - y)])
    pos = torch.tensor(pos, dtype=torch.int64)
    n, hr = pos.shape[0], 4 * tile
    ypred = torch.rand(n, 1, hr, hr, generator=g) * 70.0 - 5.0         # some negatives: exercises the clamp (:173)
    ypred[:, :, ::5, ::7] = (torch.randint(0, 600, (n, 1, (hr + 4) // 5, (hr + 6) // 7), generator=g).float() + 0.5) / 10.0   # exact .5 ties after *10
    logits = torch.randn(n, chans_build, hr, hr, generator=g) * 3.0
    return ypred, logits, pos, lr_w, lr_h
