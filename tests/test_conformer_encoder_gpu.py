"""GPU parity of the HIP Conformer encoder forward against the PyTorch-CPU float32 oracle
(oracle/conformer_oracle.py) with identical weights and inputs.

Tolerance.  Matmul inputs are bf16 (8-bit mantissa) on the device, float32 in the oracle; every layer ends in a
LayerNorm, so activations are O(1) and errors do not grow with depth.  The bar written here: relative RMS error
<= 2e-2 and max abs error <= 0.15 on the (LayerNormed, unit-scale) encoder output; the "loss curve within 1e-4"
of the north star is a float32-vs-float32 statement and belongs to the training rows."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _pair(num_blocks, seed=0, cmvn=False):
    import torch

    from mindaudio_amd.models import ConformerEncoder
    from oracle import conformer_oracle as C

    torch.manual_seed(seed)
    kw = {}
    mean = istd = None
    if cmvn:
        mean = torch.randn(80) * 0.5
        istd = torch.rand(80) + 0.5
        kw = dict(cmvn_mean=mean, cmvn_istd=istd)
    ref = C.ConformerEncoder(80, 256, 4, 2048, num_blocks, **kw).eval()
    # non-trivial BatchNorm running statistics and LayerNorm affine parameters
    with torch.no_grad():
        for m in ref.modules():
            if isinstance(m, torch.nn.BatchNorm1d):
                m.running_mean.normal_(0, 0.2)
                m.running_var.uniform_(0.5, 1.5)
                m.weight.uniform_(0.8, 1.2)
                m.bias.normal_(0, 0.1)
            if isinstance(m, C.LayerNorm):
                m.gamma.uniform_(0.8, 1.2)
                m.beta.normal_(0, 0.1)
    dut = ConformerEncoder(80, 256, 4, 2048, num_blocks, global_cmvn=(mean, istd) if cmvn else None).eval()
    missing, unexpected = dut.load_state_dict(ref.state_dict(), strict=False)
    assert not [k for k in missing if "cmvn" not in k] and not unexpected, (missing, unexpected)
    return ref, dut.cuda().prepare()


@pytest.mark.parametrize("blocks,b,tlen,cmvn", [(1, 2, 131, False), (2, 3, 203, True)])
def test_encoder_matches_oracle(blocks, b, tlen, cmvn):
    import torch

    from oracle import conformer_oracle as C

    ref, dut = _pair(blocks, seed=blocks, cmvn=cmvn)
    g = torch.Generator().manual_seed(5)
    xs = torch.randn(b, tlen, 80, generator=g)
    lens = [tlen, tlen - 40, tlen // 2][:b]
    mask = torch.zeros(b, 1, tlen)
    for i, n in enumerate(lens):
        mask[i, 0, :n] = 1
    sub = C.subsample_mask(mask)
    with torch.no_grad():
        want, _ = ref(xs, sub)
    got, m2 = dut(xs.cuda(), sub.cuda())
    got = got.cpu()
    assert got.shape == want.shape and m2.shape == sub.shape
    err = (got - want)
    rel_rms = float(err.pow(2).mean().sqrt() / want.pow(2).mean().sqrt())
    assert rel_rms <= 2e-2, rel_rms
    assert float(err.abs().max()) <= 0.15, float(err.abs().max())


def test_encoder_full_config_shapes_and_determinism():
    import torch

    from oracle import conformer_oracle as C

    ref, dut = _pair(12, seed=3)
    xs = torch.randn(2, 1000, 80, generator=torch.Generator().manual_seed(9))
    sub = C.subsample_mask(torch.ones(2, 1, 1000))
    with torch.no_grad():
        want, _ = ref(xs, sub)
    got, _ = dut(xs.cuda(), sub.cuda())
    got2, _ = dut(xs.cuda(), sub.cuda())
    assert tuple(got.shape) == (2, 249, 256)
    assert torch.equal(got, got2)  # no atomics / no run-to-run variation
    err = got.cpu() - want
    assert float(err.pow(2).mean().sqrt() / want.pow(2).mean().sqrt()) <= 2e-2
    with pytest.raises(NotImplementedError):
        dut.train()(xs.cuda(), sub.cuda())
