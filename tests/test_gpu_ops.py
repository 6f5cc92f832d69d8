"""Single-call operators (demfi_amd/ops.py: SepConvGRU, FGAC) on a real MI355X against the oracle's restatement of the same
reference modules (oracle/demfi_oracle.py: sep_conv_gru = DeMFInet.py:838-857, fgac = DeMFInet.py:386-452)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from demfi_amd import synthetic_state_dict            # noqa: E402
from demfi_amd.ops import FGAC, SepConvGRU             # noqa: E402
from oracle import demfi_oracle as O                   # noqa: E402

DEV = 'cuda:0'


@pytest.fixture(scope='module')
def sd():
    return synthetic_state_dict(0)


# fp32: summation order only.  fp16: storage of every intermediate in fp16 (z, r*h, h between the two steps; hidden layers of the gate)
TOL = {torch.float32: 2e-5, torch.float16: 6e-3}


@pytest.mark.parametrize('dtype', [torch.float32, torch.float16])
@pytest.mark.parametrize('B,H,W', [(1, 32, 64), (2, 37, 75), (3, 8, 32)])
def test_sep_conv_gru_matches_the_oracle(sd, dtype, B, H, W):
    torch.manual_seed(3)
    h = torch.tanh(torch.randn(B, 64, H, W))
    x = torch.relu(torch.randn(B, 64, H, W))
    if dtype == torch.float16:
        h, x = h.half().float(), x.half().float()        # the oracle sees the values the kernels see
    op = SepConvGRU(64, 64, dtype=dtype, device=DEV).load_state_dict(sd, prefix='Booster_Module.GB.')
    got = op(h.to(DEV), x.to(DEV))
    assert got.shape == h.shape and got.dtype == h.dtype
    want = O.sep_conv_gru(sd, h, x)
    assert (got.cpu() - want).abs().max() < TOL[dtype]
    # a second call on the cached plan, other data: no state left behind
    h2 = torch.tanh(torch.randn(B, 64, H, W)).half().float()
    keep = got.clone()
    got2 = op(h2.to(DEV), x.to(DEV))
    assert (got2.cpu() - O.sep_conv_gru(sd, h2, x)).abs().max() < TOL[dtype]
    # results are OWNED tensors (the reference nn.Module returns fresh ones): the second call on the cached plan must not
    # overwrite the first result, whatever the input dtype
    assert got.data_ptr() != got2.data_ptr() and torch.equal(got, keep)
    if dtype == torch.float16:
        g16a = op(h.to(DEV).half(), x.to(DEV).half())
        k16 = g16a.clone()
        g16b = op(h2.to(DEV).half(), x.to(DEV).half())
        assert g16a.dtype == torch.float16 and torch.equal(g16a, k16) and not torch.equal(g16a, g16b)


@pytest.mark.parametrize('dtype', [torch.float32, torch.float16])
@pytest.mark.parametrize('B,H,W', [(1, 32, 64), (2, 24, 40)])
def test_fgac_matches_the_oracle(sd, dtype, B, H, W):
    torch.manual_seed(5)
    ref = torch.randn(B, 64, H, W) * 0.5
    src = torch.randn(B, 64, H, W) * 0.5
    # ABSOLUTE sampling coordinates: inside the frame, on its border and beyond it
    ys, xs = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing='ij')
    flow = torch.stack([xs, ys])[None].repeat(B, 1, 1, 1) + torch.randn(B, 2, H, W) * 3.0
    flow[:, :, :2] -= 6.0
    if dtype == torch.float16:
        ref, src = ref.half().float(), src.half().float()
    name = 'FAC_FB_Module.shared_FGAC'
    op = FGAC(dtype=dtype, device=DEV).load_state_dict(sd, prefix=name + '.')
    out, w, diff = op(ref.to(DEV), src.to(DEV), flow.to(DEV))
    want, w_want = O.fgac(sd, name, ref, src, flow)
    assert out.shape == ref.shape and w.shape == (B, 1, H, W) and diff.shape == (B, 1, H, W)
    assert (w.cpu() - w_want).abs().max() < TOL[dtype]
    assert (out.cpu() - want).abs().max() < TOL[dtype]
    d = (want - src).abs().mean(1, keepdim=True).view(B, -1)
    d = (d - d.min(1, keepdim=True)[0])
    d = (d / d.max(1, keepdim=True)[0]).view(B, 1, H, W)
    assert (diff.cpu() - d).abs().max() < 50 * TOL[dtype]      # min-max normalisation divides by a small range
    assert float(diff.min()) == 0.0 and abs(float(diff.max()) - 1.0) < 1e-6
    # owned results: a second call (other data) leaves the first call's tensors alone
    ko, kw = out.clone(), w.clone()
    out2, w2, _ = op(src.to(DEV), ref.to(DEV), flow.to(DEV))
    assert torch.equal(out, ko) and torch.equal(w, kw) and not torch.equal(out, out2)


def test_operators_reject_what_the_reference_modules_would():
    op = SepConvGRU(device=DEV)
    with pytest.raises(RuntimeError):
        op(torch.zeros(1, 64, 8, 8, device=DEV), torch.zeros(1, 64, 8, 8, device=DEV))
    sd = synthetic_state_dict(0)
    op.load_state_dict(sd, prefix='Booster_Module.GB.')
    with pytest.raises(ValueError):
        op(torch.zeros(1, 32, 8, 8, device=DEV), torch.zeros(1, 64, 8, 8, device=DEV))       # channel mismatch
    with pytest.raises(ValueError):
        op(torch.zeros(1, 64, 8, 8, device=DEV), torch.zeros(1, 64, 8, 16, device=DEV))       # size mismatch
    with pytest.raises(ValueError):
        op(torch.zeros(1, 64, 8, 8), torch.zeros(1, 64, 8, 8))                                 # wrong device
