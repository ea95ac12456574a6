"""GPU: the headline parity claim over MANY draws (round-4 VERDICT "make the parity claim robust").

The 23-block `RRDBNet.forward_feature` (reference SR/rrdbnet_arch.py:225-240) on the fp16-operand / fp32-accumulate fast path
against the CPU oracle (oracle/srbh_oracle.py, pinned bit-identical to the imported reference by tests/golden/g4_*): 8 weight
seeds x {init, stress} x 2 input seeds = 32 whole 64 x 256 x 256 feature maps.  Tolerance: rel-L2 <= 1e-3 (BASELINE.json
north_star) on EVERY draw; the distribution is printed, and a draw above 9e-4 is reported as thin margin (the fix for one is the
dominant layer's rounding, not this number).  No trained checkpoint exists offline: all weights are seeded random
draws of the reference initialisers ('init') or of the wider 'stress' distribution (random biases)."""
import pytest
import torch

from oracle import srbh_oracle as O
from oracle import synth

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
REL_TOL = 1e-3
WEIGHT_SEEDS = (1337, 1, 2, 3, 5, 8, 13, 21)
INPUT_SEEDS = (1337, 77)


def test_forward_feature_parity_over_weight_and_input_seeds():
    from srbh_amd.rrdbnet import RRDBNet
    torch.set_num_threads(min(32, torch.get_num_threads()))
    x = torch.cat([synth.tiles(1, 8, 64, seed=s)[:, :3] for s in INPUT_SEEDS]).contiguous()
    net = RRDBNet(3, 3).to(DEV).eval()
    errs = {}
    for mode in ("init", "stress"):
        for ws in WEIGHT_SEEDS:
            sd = synth.rrdbnet_state_dict(seed=ws, mode=mode)
            net.load_state_dict(sd, strict=True)
            with torch.no_grad():
                y = net.forward_feature(x.to(DEV)).float().cpu()
            net.check_status()
            want = O.rrdbnet_forward_feature(sd, x)
            for i, s in enumerate(INPUT_SEEDS):
                errs[(mode, ws, s)] = O.rel_l2(y[i:i + 1], want[i:i + 1])
    vals = sorted(errs.values())
    worst = max(errs, key=errs.get)
    print(f"\nparity sweep: {len(vals)} draws, rel-L2 min {vals[0]:.3e} median {vals[len(vals) // 2]:.3e} max {vals[-1]:.3e} "
          f"(worst draw: mode={worst[0]} weight seed={worst[1]} input seed={worst[2]})")
    for mode in ("init", "stress"):
        v = sorted(e for k, e in errs.items() if k[0] == mode)
        print(f"  {mode:6s}: " + " ".join(f"{e:.2e}" for e in v))
    thin = {k: e for k, e in errs.items() if e > 9e-4}
    if thin:
        print(f"  THIN MARGIN (> 9e-4): {thin}")
    assert vals[-1] <= REL_TOL, f"worst draw {worst}: rel-L2 {vals[-1]:.3e} > {REL_TOL}"
