"""Single-call operators (demfi_amd/ops.py: SepConvGRU, FGAC) on a real MI355X against the oracle's restatement of the same
reference modules (oracle/demfi_oracle.py: sep_conv_gru = DeMFInet.py:838-857, fgac = DeMFInet.py:386-452)."""
import ctypes as C

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


# ------------------------------------------------------------------------------------------------------
# the same two operators through the C ABI alone (operator contexts, ABI v7): what a non-Python host binds
# ------------------------------------------------------------------------------------------------------
class _COperator:
    """ctypes-only driver of an operator context: create, load the reference module's keys, bind a caller-owned workspace, fill the
    named buffers, demfi_operator_run, read the named outputs.  torch only provides the device memory."""

    def __init__(self, kind, B, H, W, dtype, sd, prefix):
        from demfi_amd import _lib as L
        self.L, self.lib = L, L.load()
        self.ctx = C.c_void_p()
        create = self.lib.demfi_gru_sep_create if kind == 'gru' else self.lib.demfi_fgac_create
        L.check(create(B, H, W, L.F32 if dtype == torch.float32 else L.F16, C.byref(self.ctx)), 'create')
        keys = ['conv%s%d' % (g, i) for i in (1, 2) for g in 'zrq'] if kind == 'gru' else ['conv_ref_k', 'conv_source_k', 'fusion', 'w_gen', 'w_gen_2']
        for k in keys:
            for part in ('weight', 'bias'):
                t = sd[prefix + k + '.' + part].detach().float().contiguous()
                shape = (C.c_int64 * t.dim())(*t.shape)
                L.check(self.lib.demfi_load_weight(self.ctx, (k + '.' + part).encode(), t.data_ptr(), shape, t.dim()), 'load_weight')
        n = self.lib.demfi_ctx_workspace_bytes(self.ctx)
        self.ws = torch.zeros(n, dtype=torch.uint8, device=DEV)
        L.check(self.lib.demfi_ctx_bind(self.ctx, self.ws.data_ptr(), n, 0, torch.cuda.current_stream().cuda_stream), 'bind')
        self.dtype = dtype

    def buf(self, name):
        off, kind, dims = C.c_int64(0), C.c_int32(0), (C.c_int32 * 4)()
        self.L.check(self.lib.demfi_ctx_buffer(self.ctx, 0, -1, name.encode(), C.byref(off), C.byref(kind), dims), 'buffer')
        d = list(dims)
        if kind.value == 0:
            n = d[0] * d[1] * d[2] * d[3] * (4 if self.dtype == torch.float32 else 2)
            return self.ws[off.value:off.value + n].view(self.dtype).view(d[0], d[1], d[2], d[3])
        return self.ws[off.value:off.value + d[0] * d[1] * d[2] * 4].view(torch.float32).view(d[0], d[1], d[2])

    def run(self):
        self.L.check(self.lib.demfi_operator_run(self.ctx, torch.cuda.current_stream().cuda_stream), 'operator_run')
        torch.cuda.synchronize()

    def close(self):
        self.lib.demfi_ctx_destroy(self.ctx)


@pytest.mark.parametrize('dtype', [torch.float32, torch.float16])
def test_c_level_sep_conv_gru_matches_the_oracle(sd, dtype):
    """VERDICT r4 missing #3 / SURVEY 8b: demfi_gru_sep as a C entry point (DeMFInet.py:838-857)."""
    B, H, W = 2, 37, 75
    torch.manual_seed(11)
    h = torch.tanh(torch.randn(B, 64, H, W)).half().float()
    x = torch.relu(torch.randn(B, 64, H, W)).half().float()
    op = _COperator('gru', B, H, W, dtype, sd, 'Booster_Module.GB.')
    op.buf('h').copy_(h.permute(0, 2, 3, 1))
    op.buf('x').copy_(x.permute(0, 2, 3, 1))
    op.run()
    got = op.buf('out').permute(0, 3, 1, 2).float().cpu()
    assert (got - O.sep_conv_gru(sd, h, x)).abs().max() < TOL[dtype]
    assert op.lib.demfi_ctx_num_ops(op.ctx, 0, 0, 0, 0) == 4           # z | r and q per direction
    op.close()


@pytest.mark.parametrize('dtype', [torch.float32, torch.float16])
def test_c_level_fgac_matches_the_oracle(sd, dtype):
    """demfi_fgac as a C entry point (DeMFInet.py:386-452 at rr = sr = 0): out and the gate w_sr."""
    B, H, W = 2, 24, 40
    torch.manual_seed(5)
    ref = (torch.randn(B, 64, H, W) * 0.5).half().float()
    src = (torch.randn(B, 64, H, W) * 0.5).half().float()
    ys, xs = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing='ij')
    flow = torch.stack([xs, ys])[None].repeat(B, 1, 1, 1) + torch.randn(B, 2, H, W) * 3.0
    flow[:, :, :2] -= 6.0
    name = 'FAC_FB_Module.shared_FGAC'
    op = _COperator('fgac', B, H, W, dtype, sd, name + '.')
    op.buf('ref').copy_(ref.permute(0, 2, 3, 1))
    op.buf('source').copy_(src.permute(0, 2, 3, 1))
    op.buf('flow').copy_(flow.reshape(2 * B, H, W))
    op.run()
    want, w_want = O.fgac(sd, name, ref, src, flow)
    assert (op.buf('out').permute(0, 3, 1, 2).float().cpu() - want).abs().max() < TOL[dtype]
    assert (op.buf('w').cpu().view(B, 1, H, W) - w_want).abs().max() < TOL[dtype]
    op.close()
