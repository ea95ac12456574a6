"""developer aid / evidence for BENCH's `train_step.parity`: srbh_amd.gradcheck.mixed_vs_exact on fresh nets.  usage: grad_parity.py [batch] [num_block] [seed]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

if __name__ == "__main__":
    from srbh_amd import gradcheck, synth
    from srbh_amd.harness import synthetic_batch
    from srbh_amd.models import SRRegress_Cls_feature
    from srbh_amd.rrdbnet import RRDBNet
    batch = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    nb = int(sys.argv[2]) if len(sys.argv) > 2 else 23
    seed = int(sys.argv[3]) if len(sys.argv) > 3 else 1337
    dev = torch.device("cuda", 0)
    net_hr = RRDBNet(3, 3, num_block=nb)
    net_hr.load_state_dict(synth.rrdbnet_state_dict(num_block=nb, seed=1337, mode="init"))
    torch.manual_seed(seed)
    net = SRRegress_Cls_feature("efficientnet-b4", in_channels=8, super_in=64, super_mid=16, upscale=4, isaggre=True, chans_build=7)
    print(json.dumps(gradcheck.mixed_vs_exact(net_hr.to(dev), net.to(dev), synthetic_batch(batch, 4242, dev), dev)))
