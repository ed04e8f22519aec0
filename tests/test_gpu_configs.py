"""The reference's other FNO configurations run through the same HIP path (one fused train step + eval forward,
checked against the oracle at B=1): fsi (width 128, modes 4/16/16), combustion (16 input/output channels), the combustion
surrogate (17 -> 1 channels, train_surrogate.py)."""
import pytest
import torch

from conftest import rel_l2

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name,shape_in,shape_out,modes,width", [
    ("fsi", (20, 64, 64, 3), (20, 64, 64, 3), (4, 16, 16), 128),          # configs/fsi/fno.yaml
    ("combustion", (20, 64, 64, 16), (20, 64, 64, 16), (4, 16, 16), 64),  # configs/combustion/fno.yaml
    ("controlled", (10, 64, 128, 5), (10, 64, 128, 3), (4, 12, 16), 64),  # control channels in, 3 fields out
    ("surrogate", (20, 32, 32, 17), (20, 32, 32, 1), (4, 16, 16), 64),    # configs/combustion/surrogate_model/fno.yaml: 15 + 2 -> 1
])
def test_reference_fno_configs(name, shape_in, shape_out, modes, width):
    from oracle import fno3d_oracle as O
    from realpdebench_amd.model.fno import FNO3d
    from realpdebench_amd.trainer import Trainer
    torch.manual_seed(3)
    L = 2                                     # two layers keep the CPU oracle at a few seconds; kernels are per-layer
    sd = O.init_state_dict(modes, L, width, shape_in, shape_out, seed=5)
    x, y = torch.randn(1, *shape_in), torch.randn(1, *shape_out)
    loss_ref, _, grads, _ = O.loss_and_grads(sd, x, y, modes, L, shape_in, shape_out)
    m = FNO3d(*modes, L, width, shape_in, shape_out)
    m.load_state_dict(sd)
    m = m.cuda()
    tr = Trainer(m, lr=1e-3, num_update=10)
    loss = tr.step(x.cuda(), y.cuda())
    assert abs(float(loss) - float(loss_ref)) < 2e-5 * float(loss_ref), name
    got = m.grads_as_state_dict(tr.grad)
    for k in ("fc0.weight", "spectral_convs.0.weights1", "spectral_convs.1.weights4", "convs.1.weight", "fc1.weight",
              "fc2.weight", "bns.0.weight"):
        assert rel_l2(got[k].cpu(), grads[k]) < 1e-4, (name, k)
