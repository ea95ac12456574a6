"""CPU oracle of the inference epilogue  --  TEST INFRASTRUCTURE ONLY.

Literal numpy restatement of reference predict_realesanet_feature_globe.py:156-158,172-185,195-204 (same dtypes, same
operation order).  PARITY UNPINNED: that script imports gdal/rasterio/geopandas at module level and cannot be imported
here, and the reference holds no fixture for it; the integer steps are nevertheless the reference's own numpy calls."""
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
