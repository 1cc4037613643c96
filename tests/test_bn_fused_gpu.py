"""BatchNorm (+ residual) + PReLU units whose channel is ONE workgroup: statistics + apply (forward), reduce + apply (backward,
also through the (1, 2) max-pool) in one launch from registers -- the two-launch route's additions in the same order (they agree
to an ulp of invstd / of the sums: the compiler contracts the products into other fused multiply-adds) and correct against torch.
Reference: networks/classifiers.py:37-69 (ResnetBlock), 147-163 (the 1-d model's late blocks: 195 ... 476 channels on rows of
53 ... 3 frames)."""
import pytest
import torch
import torch.nn.functional as TF

from freesound_classification_amd import functional as F, _lib

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture
def restore_conv_arith():
    mode = F.get_conv_arith()
    yield
    F.set_conv_arith(mode)

# (n, c, length): cfg 3's blocks 5 - 9 at batch 128, rows at odd 4-byte offsets, a partly filled last trip of the batch
SHAPES = [(128, 195, 53), (128, 244, 26), (128, 305, 13), (128, 381, 6), (128, 476, 3), (37, 130, 107), (5, 128, 2),
          (128, 156, 107), (3, 7, 1000)]            # 1024 threads x 4 quads (block 4); few channels, long rows


def _bn(c, seed):
    g = torch.Generator().manual_seed(seed)
    bn = torch.nn.BatchNorm2d(c)
    with torch.no_grad():
        bn.weight.copy_(torch.rand(c, generator=g) + 0.5)
        bn.bias.copy_(torch.randn(c, generator=g))
        bn.running_mean.copy_(torch.randn(c, generator=g))
    return bn


def _close(a, b, rel=2e-6):
    """max |a - b| <= rel * max |b| (both None: equal)"""
    if a is None or b is None:
        return a is b
    return float((a - b).abs().max()) <= rel * max(float(b.abs().max()), 1e-30)


def test_supported_shapes():
    lib = _lib.load()
    for n, c, w in SHAPES:
        assert lib.fsc_bn_train_act_fwd_supported(n, c, w) & 1, (n, c, w)
    assert lib.fsc_bn_train_act_fwd_supported(128, 125, 215) == 3          # (block 3, forward only: 1024 threads x 8 quads)
    for n, c, w in [(128, 64, 3446), (128, 100, 430), (128, 476, 1)]:      # big; > 8 quads per thread; hw == 1
        assert lib.fsc_bn_train_act_fwd_supported(n, c, w) == 0, (n, c, w)
    x = torch.randn(128, 100, 1, 430, device=DEV)
    f = torch.empty(100, device=DEV)
    with pytest.raises(_lib.FscError):
        F.call("fsc_bn_train_act_fwd", F.ptr(x), None, 128, 100, 430, None, None, 1e-5, 0.1, None, None, F.ptr(f), F.ptr(f), F.ptr(f),
               F.ptr(f), None, None, F.ptr(torch.empty_like(x)), None, None, F.stream_ptr())


@pytest.mark.parametrize("shape", SHAPES + [(128, 125, 215)])
@pytest.mark.parametrize("with_alpha,with_res", [(False, False), (True, False), (True, True)])
def test_forward_one_launch_equals_two(shape, with_alpha, with_res, restore_conv_arith):
    F.set_conv_arith("bf16")
    n, c, w = shape
    torch.manual_seed(n + c + w)
    x = (torch.randn(n, c, 1, w) * 2 + 3)
    x[:, 3] = 7.5                                                   # a constant channel (variance 0)
    res = torch.randn(n, c, 1, w) if with_res else None
    alpha = (torch.rand(c) * 0.3 + 0.1).to(DEV) if with_alpha else None
    xd, rd = x.to(DEV), res.to(DEV) if with_res else None
    out = []
    for lazy in (False, True):
        bn = _bn(c, 5).to(DEV).train()
        st = F.bn_prepare(xd, bn, True, lazy=lazy)
        assert (st.pending is not None) == lazy
        y = F.bn_act_forward(xd, st, alpha, rd)
        assert st.pending is None
        out.append((y, st.mean, st.invstd, st.scale, st.shift, st.minmax, bn.running_mean.clone(), bn.running_var.clone()))
    for i, (a, b) in enumerate(zip(*out)):
        assert _close(a, b), i
    assert torch.equal(out[0][5], out[1][5])                                              # min / max: bit for bit
    # and against torch
    ref_bn = _bn(c, 5).train()
    z = ref_bn(x)
    if with_res:
        z = z + res
    ref = TF.prelu(z, alpha.cpu()) if with_alpha else z
    assert float((out[1][0].cpu() - ref.detach()).abs().max()) < 3e-5
    assert float((out[1][6].cpu() - ref_bn.running_mean).abs().max()) < 1e-5
    assert float((out[1][7].cpu() - ref_bn.running_var).abs().max()) < 1e-4


def test_lazy_statistics_reach_other_consumers(restore_conv_arith):
    """A lazy state handed to a consumer that is not the plain apply pass: the statistics call is issued there."""
    F.set_conv_arith("bf16")
    torch.manual_seed(0)
    x = torch.randn(128, 195, 1, 53, device=DEV)
    res = torch.randn_like(x)
    alpha = torch.full((195,), 0.25, device=DEV)
    bn1, bn2 = _bn(195, 1).to(DEV).train(), _bn(195, 1).to(DEV).train()
    st = F.bn_prepare(x, bn1, True, lazy=True)
    assert st.pending is not None
    y, feat, fidx = F.bn_act_forward_rec(x, st, alpha, res, True, True)
    assert st.pending is None
    F._PRESTATS.clear()
    st2 = F.bn_prepare(x, bn2, True)
    y2 = F.bn_act_forward(x, st2, alpha, res)
    assert torch.equal(y, y2) and torch.equal(st.mean, st2.mean) and torch.equal(feat, y2.amax(dim=(2, 3)))      # (both: the two-launch statistics)


@pytest.mark.parametrize("shape", SHAPES[:8] + [(128, 125, 215), (2, 9, 248)])
def test_forward_with_the_global_max(shape, restore_conv_arith):
    """The block-end unit (bn3 + residual + PReLU) that also serves the head's global max-pool: values and FIRST positions as
    fsc_global_maxpool_fwd finds them on the stored output (ties: a constant channel, a clamped one; a NaN wins)."""
    F.set_conv_arith("bf16")
    n, c, w = shape
    assert _lib.load().fsc_bn_train_act_fwd_supported(n, c, w) == 3
    torch.manual_seed(n + c + 3 * w)
    x = torch.randn(n, c, 1, w)
    x[:, 1] = 2.0                                       # constant channel: every position ties, index 0 wins
    res = torch.randn(n, c, 1, w)
    res[:, 1] = 0.0
    x[0, 2, 0, w // 2] = float("nan")                   # (poisons channel 2's statistics: every value of it is NaN, index 0 wins)
    alpha = (torch.rand(c) * 0.3 + 0.1).to(DEV)
    xd, rd = x.to(DEV), res.to(DEV)
    bn = _bn(c, 5).to(DEV).train()
    st = F.bn_prepare(xd, bn, True, lazy=True)
    y, feat, fidx = F.bn_act_forward_unit(xd, st, alpha, rd, True)
    assert st.pending is None
    ref_feat, ref_idx = F.global_maxpool_forward(y)
    assert torch.equal(fidx, ref_idx)
    assert torch.equal(feat.view(torch.int32), ref_feat.view(torch.int32))          # (bit for bit: NaNs and zeros included)
    bn2 = _bn(c, 5).to(DEV).train()
    y2 = F.bn_act_forward(xd, F.bn_prepare(xd, bn2, True), alpha, rd)
    ok = ~torch.isnan(y2)
    assert torch.equal(torch.isnan(y), torch.isnan(y2)) and _close(y[ok], y2[ok])
    assert int(fidx[0, 1, ]) == 0 and int(fidx[0, 2]) == 0


def test_long_planes_leave_the_global_max_to_the_record_route(restore_conv_arith):
    F.set_conv_arith("bf16")
    x = torch.randn(3, 7, 1, 1000, device=DEV)
    assert _lib.load().fsc_bn_train_act_fwd_supported(3, 7, 1000) == 1
    st = F.bn_prepare(x, _bn(7, 1).to(DEV).train(), True, lazy=True)
    assert F.bn_act_forward_unit(x, st, None, None, True) is None and st.pending is not None
    y, feat, fidx = F.bn_act_forward_unit(x, st, None, None, False)
    assert feat is None and st.pending is None and torch.isfinite(y).all()


@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("with_res,with_gmax", [(False, False), (True, False), (True, True)])
def test_backward_one_launch_equals_two(shape, with_res, with_gmax, restore_conv_arith):
    n, c, w = shape
    torch.manual_seed(n + 2 * c + w)
    x = (torch.randn(n, c, 1, w) * 2 + 3).requires_grad_()
    res = torch.randn(n, c, 1, w).requires_grad_() if with_res else None
    bn = _bn(c, 9).train()
    alpha = (torch.rand(c) * 0.3 + 0.1).requires_grad_()
    z = bn(x)
    if with_res:
        z = z + res
    y = TF.prelu(z, alpha)
    gy = torch.randn_like(y)
    g_feat = torch.randn(n, c) if with_gmax else None
    loss = (y * gy).sum()
    if with_gmax:
        loss = loss + (y.amax(dim=(2, 3)) * g_feat).sum()
    loss.backward()

    xd, rd, ad = x.detach().to(DEV), res.detach().to(DEV) if with_res else None, alpha.detach().to(DEV)
    dbn = _bn(c, 9).to(DEV).train()
    gmax = None
    got = {}
    for arith in ("f16x3", "bf16"):            # f16x3: the apply pass reports max |dx| -> two launches; bf16: one
        F.set_conv_arith(arith)
        st = F.bn_prepare(xd, dbn, True)
        if with_gmax:
            yd = F.bn_act_forward(xd, st, ad, rd)
            feat, fidx = F.global_maxpool_forward(yd)
            gmax = (g_feat.to(DEV), fidx)
        r = F.bn_act_backward(gy.to(DEV), xd, st, dbn, ad, rd, gmax=gmax, want_dres=with_res, want_chan_sum=True,
                              with_amax=True)
        assert (r[6] is not None) == (arith == "f16x3")
        got[arith] = r[:6]
    for i, (a, b) in enumerate(zip(got["f16x3"][:5], got["bf16"][:5])):
        assert _close(a, b), i
    # (the closed-form channel sum of dx is rounding residue of sums of ~1e3: compared on that scale)
    assert float((got["f16x3"][5] - got["bf16"][5]).abs().max()) < 1e-3
    dx, dres, dg, db, dal, csum = got["bf16"]
    assert float((dx.cpu() - x.grad).abs().max()) < 5e-5
    assert float((dg.cpu() - bn.weight.grad).abs().max()) < 3e-4 * max(1.0, float(bn.weight.grad.abs().max()))
    assert float((db.cpu() - bn.bias.grad).abs().max()) < 3e-4 * max(1.0, float(bn.bias.grad.abs().max()))
    assert float((dal.cpu() - alpha.grad).abs().max()) < 3e-4 * max(1.0, float(alpha.grad.abs().max()))
    if with_res:
        assert float((dres.cpu() - res.grad).abs().max()) < 2e-5


@pytest.mark.parametrize("shape", [(128, 195, 107), (128, 244, 53), (128, 476, 7), (128, 381, 12), (9, 130, 214), (128, 156, 215)])
def test_backward_through_the_pool_one_launch_equals_two(shape, restore_conv_arith):
    n, c, w = shape                                      # (w: the un-pooled row; odd: a last column the pool never read)
    torch.manual_seed(n + c + w)
    full = torch.randn(n, c, 1, w, requires_grad=True)
    bn = _bn(c, 11).train()
    alpha = (torch.rand(c) * 0.3 + 0.1).requires_grad_()
    y = TF.prelu(bn(TF.max_pool2d(full, (1, 2), (1, 2))), alpha)
    gy = torch.randn_like(y)
    y.backward(gy)
    dbn = _bn(c, 11).to(DEV).train()
    pd, pidx = F.maxpool_forward(full.detach().to(DEV), 1)
    assert _lib.load().fsc_bn_train_act_fwd_supported(n, c, w // 2) & 1
    got = {}
    for arith in ("f16x3", "bf16"):
        F.set_conv_arith(arith)
        st = F.bn_prepare(pd, dbn, True)
        r = F.bn_act_backward_unpool(gy.to(DEV), pd, st, dbn, alpha.detach().to(DEV), pidx, (n, c, 1, w), 1)
        assert (r[5] is not None) == (arith == "f16x3")
        got[arith] = r[:5]
    for i, (a, b) in enumerate(zip(got["f16x3"][:4], got["bf16"][:4])):
        assert _close(a, b), i
    assert float((got["f16x3"][4] - got["bf16"][4]).abs().max()) < 1e-3
    dc, dg, db, dal, csum = got["bf16"]
    assert float((dc.cpu() - full.grad).abs().max()) < 5e-5
    assert float((dg.cpu() - bn.weight.grad).abs().max()) < 3e-4 * max(1.0, float(bn.weight.grad.abs().max()))
    assert float((dal.cpu() - alpha.grad).abs().max()) < 3e-4 * max(1.0, float(alpha.grad.abs().max()))
