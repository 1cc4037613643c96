"""Robustness of the three-limb arithmetics -- arith 10 ("f16x6", the library default and bench.py's `value`) and arith 9
("bf16x9") -- on the inputs the reference's un-stabilised LSEP (networks/losses.py:47-58: exp of raw score gaps) really
produces: per-sample gradient rows tens of binary orders apart, and +-Inf / NaN.  VERDICT r5 weak 2: these cases were asserted
for the two-limb fast mode only (tests/test_cfg2_gpu.py::test_split_fp16_*).  Here: the pre-split (L16) route the training
step takes -- fsc_l16_pack_limbs / the limb-writing BatchNorm backward producers with their DECLARED bound ->
fsc_conv_l16_fwd (forward, input gradient) and fsc_conv_l16_wgrad -- against PyTorch's fp64 convolution / autograd on the CPU.

What must hold (replaces nn.Conv2d / BatchNorm2d backward on fp32 tensors, classifiers.py:526-531, 77-81):
* a non-finite operand element never comes out as a finite number: every output F.conv2d makes non-finite is non-finite here;
* bf16x9 (no scale): every sample row keeps fp32 RELATIVE accuracy whatever its magnitude;
* f16x6 (one power-of-two scale per tensor): rows within 2^-10 of the tensor's maximum keep fp32 relative accuracy, every row
  an ABSOLUTE error of a few fp32 ulps of the LARGEST row's products (what the fp32 sums over the batch downstream -- weight
  gradient, BatchNorm statistics -- lose anyway).  The per-row relative errors are written to the report.
"""
import pytest
import torch

pytestmark = pytest.mark.gpu

if not torch.cuda.is_available():
    pytest.skip("needs an MI355X", allow_module_level=True)

import torch.nn.functional as TF  # noqa: E402

from freesound_classification_amd import functional as F  # noqa: E402
from test_cfg2_gpu import _report  # noqa: E402

DEV = torch.device("cuda:0")
EPS = 2.0 ** -23


@pytest.fixture(params=[9, 10], ids=["bf16x9", "f16x6"])
def l3(request):
    mode0 = F.get_conv_arith()
    F.set_conv_arith(request.param)
    yield request.param
    F.set_conv_arith(mode0)


def _supported(n, c_in, c_out, h, w, k, arith):
    d = F._desc(n, c_in, c_out, h, w, k, k, arith)
    return F.conv_l16_supported(d, 0) and F.conv_l16_supported(d, 1) and F.conv_l16_wgrad_supported(d)


def _batch(c_in, c_out, h, w, k, arith):
    """Smallest batch at which the shape has a three-limb tiling in all three directions (the persistent kernels want >= 128
    work items)."""
    for n in (16, 32, 64, 128):
        if _supported(n, c_in, c_out, h, w, k, arith):
            return n
    raise AssertionError("no three-limb tiling for %s" % ((c_in, c_out, h, w, k),))


@pytest.mark.parametrize("case", [(100, 150, 16, 43, 3), (150, 150, 8, 26, 1), (64, 48, 8, 30, 3)])
def test_three_limb_gradient_rows_spread_over_2_to_minus_40(case, l3):
    c_in, c_out, h, w, k = case
    n = _batch(c_in, c_out, h, w, k, l3)
    pad = k // 2
    torch.manual_seed(sum(case))
    x = torch.randn(n, c_in, h, w)
    wt = torch.randn(c_out, c_in, k, k) / (c_in * k * k) ** 0.5
    expo = torch.linspace(0, -40, n).round()
    gy = torch.randn(n, c_out, h, w) * torch.exp2(expo).view(n, 1, 1, 1)
    dx64 = torch.nn.grad.conv2d_input(x.shape, wt.double(), gy.double(), padding=pad)
    dw64 = torch.nn.grad.conv2d_weight(x.double(), wt.shape, gy.double(), padding=pad)
    dw32 = torch.nn.grad.conv2d_weight(x, wt.shape, gy, padding=pad)
    x16, g16 = F.l16_pack(x.to(DEV)), F.l16_pack(gy.to(DEV))
    dx = F.conv_l16(g16, wt.to(DEV), None, dgrad=True).cpu()
    dw = F.conv_l16_wgrad(x16, g16, wt.shape).cpu()
    top = float(dx64.abs().max())
    e_abs = float((dx.double() - dx64).abs().max())
    assert e_abs < 8 * EPS * top, (e_abs, top)
    rel = []
    for r in range(n):
        rel.append(float((dx[r].double() - dx64[r]).abs().max()) / float(dx64[r].abs().max()))
        if l3 == 9 or expo[r] >= -10:           # exact limbs: every row; scaled limbs: rows within 2^-10 of the maximum
            assert rel[-1] < 16 * EPS, (r, float(expo[r]), rel[-1])
    _report("arith %d row-scaled dgrad %s: abs err %.2e of max %.2e; per-row rel err at 2^[%s] = [%s]" % (
        l3, case, e_abs, top, ", ".join("%d" % e for e in expo.tolist()), ", ".join("%.1e" % v for v in rel)))
    e_dw = float((dw.double() - dw64).abs().max())
    e_dw32 = float((dw32.double() - dw64).abs().max())
    assert e_dw < 4.0 * e_dw32 + 8 * EPS * float(dw64.abs().max()), (e_dw, e_dw32)


def _bn(c, gen):
    bn = torch.nn.BatchNorm2d(c)
    with torch.no_grad():
        bn.weight.copy_(1.0 + 0.3 * torch.randn(c, generator=gen))
        bn.bias.copy_(0.2 * torch.randn(c, generator=gen))
    return bn.train()


def _bn_prelu_backward_cpu(x, bn, alpha, dy, dtype):
    """dx of PReLU(BatchNorm(x)) in train mode by autograd, in `dtype`."""
    xx = x.to(dtype).requires_grad_(True)
    y = TF.batch_norm(xx, None, None, bn.weight.detach().to(dtype), bn.bias.detach().to(dtype), True, 0.0, bn.eps)
    y = TF.prelu(y, alpha.to(dtype))
    y.backward(dy.to(dtype))
    return xx.grad.detach()


def test_lsep_shaped_gradient_through_the_limb_writing_bn_backward_and_dgrad(l3):
    """The chain of a residual unit's backward: upstream gradient dy whose per-sample rows are 2^0 ... 2^-30 apart (LSEP:
    one sample far off its labels, the others nearly satisfied) -> BatchNorm + PReLU backward, which WRITES the limbs of its
    result under the bound it declares from its reduce pass (fsc_bn_act_bwd, FSC_BN_L16_*) -> 3x3 input-gradient
    convolution on those limbs -> weight gradient.  Against the same chain in fp64 on the CPU; per-sample relative errors reported
    beside those of the chain in PyTorch fp32."""
    c, c_prev, h, w, k = 100, 100, 16, 43, 3
    n = _batch(c_prev, c, h, w, k, l3)
    gen = torch.Generator().manual_seed(11)
    x = torch.randn(n, c, h, w, generator=gen)               # the BatchNorm's input (a convolution output)
    xin = torch.randn(n, c_prev, h, w, generator=gen)        # the input of that convolution (for its weight gradient)
    bn = _bn(c, gen)
    alpha = 0.25 + 0.1 * torch.rand(c, generator=gen)
    wt = torch.randn(c, c_prev, k, k, generator=gen) / (c_prev * k * k) ** 0.5
    expo = torch.tensor([0.0, -30.0] + torch.linspace(-1, -29, n - 2).round().tolist())
    dy = torch.randn(n, c, h, w, generator=gen) * torch.exp2(expo).view(n, 1, 1, 1)
    # fp64 / fp32 chains on the CPU
    chain = {}
    for dtype in (torch.float64, torch.float32):
        dz = _bn_prelu_backward_cpu(x, bn, alpha, dy, dtype)
        chain[dtype] = (dz, torch.nn.grad.conv2d_input(xin.shape, wt.to(dtype), dz, padding=1),
                        torch.nn.grad.conv2d_weight(xin.to(dtype), wt.shape, dz, padding=1))
    dz64, dx64, dw64 = chain[torch.float64]
    dz32, dx32, dw32 = chain[torch.float32]
    # accelerated chain
    bnd = torch.nn.BatchNorm2d(c).to(DEV).train()
    bnd.load_state_dict(bn.state_dict())
    xd = x.to(DEV)
    st = F.bn_prepare(xd, bnd, True)
    res = F.bn_act_backward(dy.to(DEV), xd, st, bnd, alpha.to(DEV), want_dres=False, want_chan_sum=True, with_amax=True, l16=True,
                            want_f32=True)
    dz, t = res[0], res[-1]
    assert isinstance(t, F.L16)
    if l3 == 10:                                             # the declared bound covers the true maximum and is not wildly above it
        top = float(t.amax.max())
        assert float(dz.abs().max()) <= top <= 8.0 * float(dz.abs().max())
    dx = F.conv_l16(t, wt.to(DEV), None, dgrad=True).cpu()
    dw = F.conv_l16_wgrad(F.l16_pack(xin.to(DEV)), t, wt.shape).cpu()
    sc = float(dx64.abs().max())
    assert float((dz.cpu().double() - dz64).abs().max()) < 16 * EPS * float(dz64.abs().max())
    e = float((dx.double() - dx64).abs().max())
    e32 = float((dx32.double() - dx64).abs().max())
    assert e < max(4.0 * e32, 16 * EPS * sc), (e, e32, sc)
    rel = [float((dx[r].double() - dx64[r]).abs().max()) / float(dx64[r].abs().max()) for r in range(n)]
    rel32 = [float((dx32[r].double() - dx64[r]).abs().max()) / float(dx64[r].abs().max()) for r in range(n)]
    rowmax = [float(dx64[r].abs().max()) / sc for r in range(n)]
    _report("arith %d LSEP-shaped chain (BN backward producer -> dgrad): |dx| row maxima / tensor max = [%s]; per-sample rel err "
            "[%s]; PyTorch fp32 chain [%s]" % (l3, ", ".join("%.1e" % v for v in rowmax), ", ".join("%.1e" % v for v in rel),
                                                ", ".join("%.1e" % v for v in rel32)))
    # BatchNorm's batch statistics mix the samples: every row of dz carries the mean terms of the LARGE rows, so no row of dx is
    # more than a few binary orders below the tensor's maximum -- and every row keeps fp32 relative accuracy in both arithmetics
    for r in range(n):
        assert rel[r] < max(4.0 * rel32[r], 64 * EPS), (r, rel[r], rel32[r])
    e_dw = float((dw.double() - dw64).abs().max())
    e_dw32 = float((dw32.double() - dw64).abs().max())
    assert e_dw < 4.0 * e_dw32 + 8 * EPS * float(dw64.abs().max()), (e_dw, e_dw32)


@pytest.mark.parametrize("bad", [float("inf"), float("-inf"), float("nan")])
@pytest.mark.parametrize("case", [(100, 100, 16, 43, 3), (100, 100, 16, 43, 1)])
def test_three_limb_non_finite_operands_surface(case, bad, l3):
    """An Inf or NaN in an operand (activation, gradient or weight) must come out non-finite wherever F.conv2d's result is
    non-finite, in forward, input gradient and weight gradient -- never as a finite number (a saturated limb)."""
    c_in, c_out, h, w, k = case
    n = _batch(c_in, c_out, h, w, k, l3)
    pad = k // 2
    torch.manual_seed(1)
    x = torch.randn(n, c_in, h, w)
    wt = torch.randn(c_out, c_in, k, k) / (c_in * k * k) ** 0.5
    gy = torch.randn(n, c_out, h, w)
    wd = wt.to(DEV)
    xb = x.clone()
    xb[1, 37, 5, 11] = bad
    ref = TF.conv2d(xb, wt, None, padding=pad)
    got = F.conv_l16(F.l16_pack(xb.to(DEV)), wd, None).cpu()
    assert not torch.isfinite(got[~torch.isfinite(ref)]).any()
    ok = torch.isfinite(ref)
    if bad != bad and l3 == 9:       # NaN through exact limbs: the rest of the output is untouched and still right
        assert float((got[ok] - ref[ok]).abs().max()) < 1e-4
    elif l3 == 10 and bad == bad:    # scaled limbs: an Inf makes the declared maximum Inf -- EVERY output of the call is marked
        assert not torch.isfinite(got).any()
    assert not ((got - ref).abs()[ok & torch.isfinite(got)] > 1e-4).any()       # whatever stayed finite is right
    gb = gy.clone()
    gb[0, 3, 2, 2] = bad
    g16 = F.l16_pack(gb.to(DEV))
    rdx = torch.nn.grad.conv2d_input(x.shape, wt, gb, padding=pad)
    dx = F.conv_l16(g16, wd, None, dgrad=True).cpu()
    assert not torch.isfinite(dx[~torch.isfinite(rdx)]).any()
    rdw = torch.nn.grad.conv2d_weight(x, wt.shape, gb, padding=pad)
    dw = F.conv_l16_wgrad(F.l16_pack(x.to(DEV)), g16, wt.shape).cpu()
    assert not torch.isfinite(dw[~torch.isfinite(rdw)]).any()
    rdw2 = torch.nn.grad.conv2d_weight(xb, wt.shape, gy, padding=pad)
    dw2 = F.conv_l16_wgrad(F.l16_pack(xb.to(DEV)), F.l16_pack(gy.to(DEV)), wt.shape).cpu()
    assert not torch.isfinite(dw2[~torch.isfinite(rdw2)]).any()
    wb = wt.clone()
    wb[5, 7] = bad
    rw = TF.conv2d(x, wb, None, padding=pad)
    gw = F.conv_l16(F.l16_pack(x.to(DEV)), wb.to(DEV), None).cpu()
    assert not torch.isfinite(gw[~torch.isfinite(rw)]).any()
    rdxw = torch.nn.grad.conv2d_input(x.shape, wb, gy, padding=pad)
    dxw = F.conv_l16(F.l16_pack(gy.to(DEV)), wb.to(DEV), None, dgrad=True).cpu()
    assert not torch.isfinite(dxw[~torch.isfinite(rdxw)]).any()


@pytest.mark.parametrize("bad", [float("inf"), float("nan")])
def test_non_finite_gradient_through_the_limb_writing_bn_backward(bad, l3):
    """A non-finite upstream gradient element reaches the BatchNorm backward producer: its reduce pass sums it into the channel's
    moments, so (as in PyTorch) the whole channel of dx is non-finite -- and the bound the producer declares for the scaled
    limbs must not hide that: the input gradient and the weight gradient computed from the limbs are non-finite wherever the CPU
    chain's are."""
    c, h, w, k = 100, 16, 43, 3
    n = _batch(c, c, h, w, k, l3)
    gen = torch.Generator().manual_seed(12)
    x = torch.randn(n, c, h, w, generator=gen)
    xin = torch.randn(n, c, h, w, generator=gen)
    bn = _bn(c, gen)
    alpha = 0.25 + 0.1 * torch.rand(c, generator=gen)
    wt = torch.randn(c, c, k, k, generator=gen) / (c * k * k) ** 0.5
    dy = torch.randn(n, c, h, w, generator=gen)
    dy[2, 41, 7, 9] = bad
    dz_ref = _bn_prelu_backward_cpu(x, bn, alpha, dy, torch.float32)
    assert not torch.isfinite(dz_ref[:, 41]).any() and torch.isfinite(dz_ref[:, 40]).all()
    rdx = torch.nn.grad.conv2d_input(xin.shape, wt, dz_ref, padding=1)
    rdw = torch.nn.grad.conv2d_weight(xin, wt.shape, dz_ref, padding=1)
    bnd = torch.nn.BatchNorm2d(c).to(DEV).train()
    bnd.load_state_dict(bn.state_dict())
    xd = x.to(DEV)
    st = F.bn_prepare(xd, bnd, True)
    res = F.bn_act_backward(dy.to(DEV), xd, st, bnd, alpha.to(DEV), want_dres=False, want_chan_sum=True, with_amax=True, l16=True,
                            want_f32=True)
    dz, t = res[0].cpu(), res[-1]
    assert not torch.isfinite(dz[~torch.isfinite(dz_ref)]).any()
    back = F.l16_unpack(t).cpu()
    assert not torch.isfinite(back[~torch.isfinite(dz_ref)]).any()              # the limbs themselves carry it
    dx = F.conv_l16(t, wt.to(DEV), None, dgrad=True).cpu()
    assert not torch.isfinite(dx[~torch.isfinite(rdx)]).any()
    dw = F.conv_l16_wgrad(F.l16_pack(xin.to(DEV)), t, wt.shape).cpu()
    assert not torch.isfinite(dw[~torch.isfinite(rdw)]).any()
