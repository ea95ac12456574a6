"""GPU: srbh_amd.optim.Adam (one libsrbh launch, csrc/srbh_optim.hip) against torch.optim.Adam -- the optimizer the reference builds at
train.py:170-179 (lr, weight_decay=1e-4, a second parameter group for the loss log_vars) -- over several steps: parameters and both
moments to fp32 rounding; parameters without a gradient are skipped as torch skips them; state_dict round trip."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def _params(seed):
    g = torch.Generator().manual_seed(seed)
    shapes = [(48, 8, 3, 3), (48,), (1632, 272, 1, 1), (7,), (16, 16, 3, 3), (4099,), (1,), (3, 5)]
    return [torch.nn.Parameter((torch.randn(s, generator=g) * 0.3).to(DEV)) for s in shapes]


def test_matches_torch_adam_over_steps_with_two_groups_and_missing_grads():
    from srbh_amd.optim import Adam
    pa, pb = _params(1), _params(1)
    lva, lvb = torch.nn.Parameter(torch.zeros(1, device=DEV)), torch.nn.Parameter(torch.zeros(1, device=DEV))
    oa = Adam(pa, lr=1e-3, weight_decay=1e-4)
    oa.add_param_group({"params": [lva], "lr": 1e-3})
    ob = torch.optim.Adam(pb, lr=1e-3, weight_decay=1e-4)
    ob.add_param_group({"params": [lvb], "lr": 1e-3})
    g = torch.Generator().manual_seed(9)
    for step in range(7):
        for k, (x, y) in enumerate(zip(pa + [lva], pb + [lvb])):
            if k == 3 and step % 2 == 0:        # a parameter that gets no gradient in some steps
                x.grad = y.grad = None
                continue
            gr = (torch.randn(x.shape, generator=g) * (10.0 ** (step - 3))).to(DEV)      # gradient scales from 1e-3 to 1e3
            x.grad, y.grad = gr.clone(), gr.clone()
        oa.step()
        ob.step()
    torch.cuda.synchronize()
    for x, y in zip(pa + [lva], pb + [lvb]):
        assert _rel(x, y) <= 2e-6, (tuple(x.shape), _rel(x, y))
        sa, sb = oa.state[x], ob.state[y]
        assert _rel(sa["exp_avg"], sb["exp_avg"]) <= 2e-6 and _rel(sa["exp_avg_sq"], sb["exp_avg_sq"]) <= 2e-6
        assert float(sa["step"]) == float(sb["step"])


def test_state_dict_round_trip_with_torch_adam():
    from srbh_amd.optim import Adam
    pa, pb = _params(2), _params(2)
    oa, ob = Adam(pa, lr=2e-3, weight_decay=1e-4), torch.optim.Adam(pb, lr=2e-3, weight_decay=1e-4)
    g = torch.Generator().manual_seed(3)
    for _ in range(3):
        for x, y in zip(pa, pb):
            gr = torch.randn(x.shape, generator=g).to(DEV)
            x.grad, y.grad = gr.clone(), gr.clone()
        oa.step(); ob.step()
    # torch's state into ours, ours into torch's; two more steps each way
    oc = Adam(pa, lr=2e-3, weight_decay=1e-4)
    oc.load_state_dict(ob.state_dict())
    od = torch.optim.Adam(pb, lr=2e-3, weight_decay=1e-4)
    od.load_state_dict(oa.state_dict())
    for _ in range(2):
        for x, y in zip(pa, pb):
            gr = torch.randn(x.shape, generator=g).to(DEV)
            x.grad, y.grad = gr.clone(), gr.clone()
        oc.step(); od.step()
    torch.cuda.synchronize()
    for x, y in zip(pa, pb):
        assert _rel(x, y) <= 5e-6


def test_param_group_added_mid_training_starts_at_step_zero_like_torch():
    """round-5 ADVICE: a parameter that joins after t steps has its OWN bias corrections (step 1 at its first update), as in torch"""
    from srbh_amd.optim import Adam
    pa, pb = _params(4), _params(4)
    la, lb = torch.nn.Parameter(torch.ones(5, device=DEV)), torch.nn.Parameter(torch.ones(5, device=DEV))
    oa, ob = Adam(pa, lr=1e-3), torch.optim.Adam(pb, lr=1e-3)
    g = torch.Generator().manual_seed(5)

    def sweep(xs, ys):
        for x, y in zip(xs, ys):
            gr = torch.randn(x.shape, generator=g).to(DEV)
            x.grad, y.grad = gr.clone(), gr.clone()
        oa.step(); ob.step()

    for _ in range(5):
        sweep(pa, pb)
    oa.add_param_group({"params": [la]})
    ob.add_param_group({"params": [lb]})
    for _ in range(3):
        sweep(pa + [la], pb + [lb])
    torch.cuda.synchronize()
    assert float(oa.state[la]["step"]) == float(ob.state[lb]["step"]) == 3.0
    assert _rel(la, lb) <= 2e-6 and _rel(pa[0], pb[0]) <= 2e-6
