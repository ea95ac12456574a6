"""CPU restatement of ONE height-stage training step  --  TEST INFRASTRUCTURE ONLY (bench.py's cpu_baseline leg, tests).

train.py:133-179,243-256 of the reference: frozen RRDBNet.forward_feature under no_grad on lr[:, rgbseq]
(`srbh_oracle.rrdbnet_forward_feature`, pinned against the imported reference: tests/golden g3/g4) ->
SRRegress_Cls_feature.forward (mymodels.py:270-293: encoder -> hrfeat -> decoder1 -> aggre_height -> reg -> decoder2 ->
seg; the 256x256 head through the functional oracle `srbh_oracle.hrfeature / hrfuse_residual`, pinned by g6/g7; the
EfficientNet-B4 encoder / U-Net decoders are the stock-op restatement of the absent third-party dependency, a18,
parity unpinned) -> MSE_adapt_weight x2 + CE_DICE_adapt_weight (`loss_oracle`, pinned by g10) -> backward -> Adam(lr 1e-3,
wd 1e-4) with the three log_vars as an extra param group.  Everything fp32 on the host cores."""
import torch
import torch.nn.functional as F

from . import loss_oracle as LO
from . import srbh_oracle as O


class CpuTrainStep:
    def __init__(self, rrdb_sd, model, lr=1e-3):
        """`model`: an SRRegress_Cls_feature instance ON THE CPU -- used as the parameter container (its own forward is
        libsrbh-only and never called here) and for its stock-op encoder / decoder sub-modules."""
        self.rrdb_sd = {k: v.float() for k, v in rrdb_sd.items()}
        self.model = model.train()
        self.log_vars = [torch.nn.Parameter(torch.zeros(1)) for _ in range(3)]       # selfloss.py:84,151 (init 0)
        self.opt = torch.optim.Adam(model.parameters(), lr=lr, weight_decay=1e-4)     # train.py:170-171
        self.opt.add_param_group({"params": self.log_vars, "lr": lr})               # train.py:172-179

    def forward(self, x, fea):
        m = self.model
        sd = dict(m.state_dict(keep_vars=True))
        feats = m.encoder(x)
        sup = O.hrfeature(sd, "hrfeat.", fea, True)
        hfea = m.decoder1(*feats)
        aggre = F.conv2d(hfea, sd["aggre_height.weight"], sd["aggre_height.bias"], 1, 1)
        height = O.hrfuse_residual(sd, "reg.", hfea, sup, True)
        build = O.hrfuse_residual(sd, "seg.", m.decoder2(*feats), sup, True)
        return height, build, aggre

    def __call__(self, batch):
        lr, height, height_aggre, build, weight, weight_aggre = batch
        with torch.no_grad():
            fea = O.rrdbnet_forward_feature(self.rrdb_sd, lr[:, :3].contiguous())
        hp, bp, ap = self.forward(lr, fea)
        loss = (LO.mse_adapt_weight(hp.squeeze(1), height, weight, self.log_vars[0])
                + LO.mse_adapt_weight(ap.squeeze(1), height_aggre, weight_aggre, self.log_vars[1])
                + LO.ce_dice_adapt_weight(bp, build, weight, self.log_vars[2]))
        self.opt.zero_grad(set_to_none=True)
        loss.sum().backward()
        self.opt.step()
        return float(loss.detach().sum())
