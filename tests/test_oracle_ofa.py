"""The CPU restatement of the once-for-all net (oracle/ofa.py) against the reference's own outputs (G8)."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))


@pytest.mark.parametrize("name", ["default_k7", "default_k3", "default_k5_narrow", "small_k35"])
def test_oracle_matches_reference(golden_dir, name):
    from make_golden_ofa import CASES, choose, perturb
    from noise_amd.ofa import OFASpectralNets      # parameter containers only: the arithmetic below is the oracle's
    from oracle import ofa as oofa
    g = np.load(os.path.join(golden_dir, "g8_ofa.npz"))
    kw, seed, _, sub = CASES[name]
    torch.manual_seed(seed)
    net = OFASpectralNets(**kw)
    perturb(net, seed)
    choose(net, **sub)
    x = torch.from_numpy(g[f"{name}_x"])
    net.train()
    y1 = oofa.forward(net, x)
    y2 = oofa.forward(net, x)
    assert [net.H_padding, net.W_padding] == list(g[f"{name}_padding"])
    first_bn, deep_bn = net.encoder[0].bn.bn, net.encoder[-1].point_linear.bn.bn
    assert [int(first_bn.num_batches_tracked), int(deep_bn.num_batches_tracked)] == list(g[f"{name}_nbt"])
    # the same torch operators in the same order on the same CPU: equal up to the summation order inside them
    np.testing.assert_allclose(y1.numpy(), g[f"{name}_train"], rtol=0, atol=1e-5)
    np.testing.assert_allclose(y2.numpy(), g[f"{name}_train2"], rtol=0, atol=1e-5)
    np.testing.assert_allclose(torch.stack([deep_bn.running_mean, deep_bn.running_var]).numpy(), g[f"{name}_deep_running"],
                               rtol=1e-5, atol=1e-6)
    net.eval()
    np.testing.assert_allclose(oofa.forward(net, x).numpy(), g[f"{name}_eval"], rtol=0, atol=1e-5)
