"""The fp32-exact route (arith 9, "bf16x9") on PRE-SPLIT operands: three exact bf16 limbs per fp32 value, all nine limb
products on v_mfma_f32_16x16x32_bf16, fp32 accumulation (csrc/conv_l3.hip, conv_l16_wgrad.hip<.., 3, 9>, the three-limb
producers of norm_act.hip).  Replaces nn.Conv2d on fp32 tensors (reference networks/classifiers.py:526-531, 77-81) at the
reference's own precision: every product is the exact product of the two fp32 operands.  And the same kernels on three SCALED
fp16 limbs with six products (arith 10, "f16x6": products to 2^-32, operands exact down to 2^-16 of their tensor's maximum):
every test of the route runs in both arithmetics, against the same bounds.

* the format: fp32 -> limbs -> fp32 is the identity, bit for bit;
* known-answer products that need all nine limb terms (the six-product and the two-limb fp16 arithmetic round differently);
* every convolution of the cfg-2 model that takes the route, forward / input gradient / accumulating input gradient / weight
  gradient, with the kernel instantiation batch 128 selects (asserted), against PyTorch's fp64 convolution on the CPU;
* the producers: what the BatchNorm / PReLU kernels write as limbs equals what they write as fp32, bit for bit
  (forward, backward, backward + un-pooling);
* the fused 2 x 2 max-pool and the BatchNorm statistics of the convolution epilogue against the separate passes.
"""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu

if not torch.cuda.is_available():
    pytest.skip("needs an MI355X", allow_module_level=True)

import torch.nn.functional as TF  # noqa: E402

from freesound_classification_amd import functional as F  # noqa: E402
from test_cfg2_gpu import LAYERS, _report  # noqa: E402

DEV = torch.device("cuda:0")


@pytest.fixture
def bf16x9():
    mode0 = F.get_conv_arith()
    F.set_conv_arith(9)
    yield
    F.set_conv_arith(mode0)


@pytest.fixture(params=[9, 10], ids=["bf16x9", "f16x6"])
def l3(request):
    """The three-limb arithmetics: yields fsc_conv_desc.arith."""
    mode0 = F.get_conv_arith()
    F.set_conv_arith(request.param)
    yield request.param
    F.set_conv_arith(mode0)


SUFFIX = {9: ",9>", 10: ",6,f16>"}
FMT = {9: 3, 10: 4}


def test_three_limb_format_is_exact(bf16x9):
    """x -> (h, m, l) -> h + (m + l) reproduces every finite normal fp32 value bit for bit, whatever its magnitude (no scale,
    no declared maximum), for channel counts around the octet edges and odd plane sizes."""
    gen = torch.Generator(device=DEV).manual_seed(5)
    for n, c, hw in ((3, 8, 64), (2, 13, 77), (1, 100, 215), (4, 33, 5)):
        x = torch.randn(n, c, hw, 1, device=DEV, generator=gen)
        x = x * torch.exp2(torch.randint(-100, 100, x.shape, device=DEV, generator=gen).float())      # 2^-100 .. 2^100
        t = F.l16_pack(x)
        assert t.limbs == 3 and t.amax is None
        assert t.data.numel() * 4 == n * ((c + 7) // 8) * 3 * hw * 16
        back = F.l16_unpack(t)
        assert torch.equal(back, x)


def test_scaled_fp16_three_limb_format():
    """arith 10: x * s = h + m + l in fp16 with the tensor's power-of-two scale.  Elements within 2^-16 of the tensor's maximum come
    back bit for bit; smaller ones within 2^-39 of the maximum (the fp16 subnormal step under the scale)."""
    mode0 = F.get_conv_arith()
    F.set_conv_arith(10)
    try:
        gen = torch.Generator(device=DEV).manual_seed(6)
        for n, c, hw in ((3, 8, 64), (2, 13, 77), (1, 100, 215), (4, 33, 5)):
            for top in (-60, 0, 50):
                x = torch.randn(n, c, hw, 1, device=DEV, generator=gen).clamp_(-3.9, 3.9)
                x = x * torch.exp2(torch.randint(-30, 1, x.shape, device=DEV, generator=gen).float() + top)
                t = F.l16_pack(x)
                assert t.limbs == 4 and t.amax is not None
                assert t.data.numel() * 4 == n * ((c + 7) // 8) * 3 * hw * 16
                back = F.l16_unpack(t)
                amax = float(x.abs().max())
                big = x.abs() >= 2.0 ** -15 * amax              # (2^-16 of the power of two above the maximum)
                assert torch.equal(back[big], x[big])
                assert float((back - x).abs().max()) <= 2.0 ** -39 * amax
                assert int(big.sum()) > x.numel() // 4 and int((~big).sum()) > x.numel() // 4
    finally:
        F.set_conv_arith(mode0)


def _single_product(a, b, arith):
    """a * b through the 1x1 convolution kernels: 96 input channels of which one carries the operands."""
    n, c, h, w = 32, 96, 16, 64
    x = torch.zeros(n, c, h, w, device=DEV)
    x[:, 5] = a
    wt = torch.zeros(48, c, 1, 1, device=DEV)
    wt[7, 5, 0, 0] = b
    mode0 = F.get_conv_arith()
    try:
        F.set_conv_arith(arith)
        if arith in (9, 10):
            assert F.conv_l16_supported(F._desc(n, c, 48, h, w, 1, 1, arith), 0)
            y = F.conv_l16(F.l16_pack(x), wt, None)
        else:
            y = F.conv_forward(x, wt, None)
    finally:
        F.set_conv_arith(mode0)
    vals = y[:, 7].flatten()
    assert float((vals - vals[0]).abs().max()) == 0.0
    return float(vals[0])


def test_products_need_all_nine_limb_terms():
    """a = 1 + 2^-8 + 2^-16 has the limbs (1, 2^-8, 2^-16); a * a = 1 + 2^-7 + 3 * 2^-16 + 2^-23 + 2^-32, whose fp32 rounding keeps
    the 2^-23 that only the m * l + l * m terms supply: arith 9 returns the correctly rounded exact product (six products
    would be one ulp short)."""
    a = 1.0 + 2.0 ** -8 + 2.0 ** -16
    exact = torch.tensor(a, dtype=torch.float64) ** 2
    want = float(exact.float())
    assert want == 1.0 + 2.0 ** -7 + 3 * 2.0 ** -16 + 2.0 ** -23
    got9 = _single_product(a, a, 9)
    assert got9 == want, (got9, want)
    assert want != 1.0 + 2.0 ** -7 + 3 * 2.0 ** -16         # (what six products -- without m * l, l * m, l * l -- add up to)
    # random operands with full 24-bit significands: the nine-product result is the fp32 sum of nine EXACT terms (small first),
    # within one fp32 ulp of the exact product and never further than PyTorch's own fp32 multiply
    gen = torch.Generator().manual_seed(3)
    for _ in range(8):
        u = float(1.0 + torch.rand(1, generator=gen, dtype=torch.float64).float())
        v = float(1.0 + torch.rand(1, generator=gen, dtype=torch.float64).float())
        got = _single_product(u, v, 9)
        ex = u * v                                            # exact in fp64 (48 significant bits)
        assert abs(got - ex) <= 2.0 ** -23 * ex, (u, v, got, ex)


def test_six_scaled_fp16_products_carry_24_bit_operands():
    """a = 1 + 2^-12 + 2^-23 needs 24 significand bits: two fp16 limbs hold 1 + 2^-12 (arith 3 loses the last bit of a * 1), three hold
    a exactly (arith 10); random 24-bit operands: within one fp32 ulp of the exact product (the dropped limb pairs are <= 2^-32)."""
    a = 1.0 + 2.0 ** -12 + 2.0 ** -23
    assert _single_product(a, 1.0, 10) == a
    assert _single_product(a, 1.0, 3) == 1.0 + 2.0 ** -12
    gen = torch.Generator().manual_seed(4)
    for _ in range(8):
        u = float(1.0 + torch.rand(1, generator=gen, dtype=torch.float64).float())
        v = float(1.0 + torch.rand(1, generator=gen, dtype=torch.float64).float())
        got = _single_product(u, v, 10)
        assert abs(got - u * v) <= 2.0 ** -23 * u * v, (u, v, got, u * v)


def _plans(n, layer, arith=9):
    c_in, c_out, h, w, k = layer
    d = F._desc(n, c_in, c_out, h, w, k, k, arith)
    return [F.l16_plan_name(d, 0), F.l16_plan_name(d, 1), F.l16_wgrad_plan_name(d)]


def _batch_for(layer, arith=9):
    c_in, c_out, h, w, k = layer
    if h * w <= 8 * 26:
        return 128
    full = _plans(128, layer, arith)
    for n in (2, 4, 8, 16, 32, 64):
        d = F._desc(n, c_in, c_out, h, w, k, k, arith)
        if all(F.conv_l16_supported(d, m) for m in (0, 1)) and F.conv_l16_wgrad_supported(d) and _plans(n, layer, arith) == full:
            return n
    return 128


L3_LAYERS = [l for l in LAYERS if l[0] >= 32 and l[2] * l[3] > 2 * 6]


def test_the_cfg2_layers_take_the_three_limb_route(l3):
    """All 3 x 3 and 1 x 1 convolutions of the cfg-2 model from 100 channels @ 64 x 215 down to 759 @ 4 x 13 have a tiling in
    every direction at batch 128 (the stem and the 2 x 6-pixel block stay on the fp32-input nine-product kernels)."""
    for layer in L3_LAYERS:
        c_in, c_out, h, w, k = layer
        d = F._desc(128, c_in, c_out, h, w, k, k, l3)
        assert F.conv_l16_supported(d, 0) and F.conv_l16_supported(d, 1) and F.conv_l16_wgrad_supported(d), layer
        names = _plans(128, layer, l3)
        assert names[0].startswith("conv_l3_fwd_kernel<%d,%d," % (k, k)) and names[0].split(" ")[0].endswith(SUFFIX[l3]), names
        assert names[2].startswith("conv_l3_wgrad_kernel<%d,%d," % (k, k)) and names[2].split(" ")[0].endswith(SUFFIX[l3]), names


@pytest.mark.parametrize("layer", L3_LAYERS, ids=["%dto%d_%dx%d_k%d" % l for l in L3_LAYERS])
def test_cfg2_layer_on_three_limbs_against_fp64(layer, l3):
    c_in, c_out, h, w, k = layer
    n = _batch_for(layer, l3)
    names = _plans(n, layer, l3)
    assert names == _plans(128, layer, l3), (names, _plans(128, layer, l3))
    torch.manual_seed(c_in * 7 + c_out + h)
    pad = k // 2
    x = torch.randn(n, c_in, h, w)
    wt = torch.randn(c_out, c_in, k, k) / (c_in * k * k) ** 0.5
    b = torch.randn(c_out)
    gy = torch.randn(n, c_out, h, w)
    y64 = TF.conv2d(x.double(), wt.double(), b.double(), padding=pad)
    dx64 = torch.nn.grad.conv2d_input(x.shape, wt.double(), gy.double(), padding=pad)
    dw64 = torch.nn.grad.conv2d_weight(x.double(), wt.shape, gy.double(), padding=pad)
    e_y = float((TF.conv2d(x, wt, b, padding=pad).double() - y64).abs().max())
    e_dx = float((torch.nn.grad.conv2d_input(x.shape, wt, gy, padding=pad).double() - dx64).abs().max())
    e_dw = float((torch.nn.grad.conv2d_weight(x, wt.shape, gy, padding=pad).double() - dw64).abs().max())
    xd, wd, bd, gd = x.to(DEV), wt.to(DEV), b.to(DEV), gy.to(DEV)
    x16, g16 = F.l16_pack(xd), F.l16_pack(gd)
    y = F.conv_l16(x16, wd, bd).cpu()
    dx = F.conv_l16(g16, wd, None, dgrad=True).cpu()
    base = torch.randn_like(x)
    dxa = F.conv_l16(g16, wd, None, dgrad=True, accumulate_into=base.to(DEV)).cpu()
    dw = F.conv_l16_wgrad(x16, g16, wt.shape).cpu()
    g_y = float((y.double() - y64).abs().max())
    g_dx = float((dx.double() - dx64).abs().max())
    g_dxa = float((dxa.double() - (dx64 + base.double())).abs().max())
    g_dw = float((dw.double() - dw64).abs().max())
    _report("%-22s arith %2d (three limbs) n %3d  fwd %.2e (torch f32 %.2e, x%.2f) %s | dgrad %.2e (%.2e, x%.2f) %s | wgrad %.2e (%.2e, x%.2f) %s"
            % ("%dto%d_%dx%d_k%d" % layer, l3, n, g_y, e_y, g_y / e_y, names[0], g_dx, e_dx, g_dx / e_dx, names[1], g_dw, e_dw,
               g_dw / e_dw, names[2]))
    eps = 2.0 ** -23

    def bound(e32, k_terms, ref):            # test_cfg2_gpu.py: 5x PyTorch's own fp32 error, or one serial fp32 chain of K / 32 blocks
        return max(5.0 * e32, 2.0 * eps * (k_terms / 4.0) ** 0.5 * float(ref.abs().max())) + 1e-7

    assert g_y < bound(e_y, c_in * k * k, y64), (g_y, e_y)
    assert g_dx < bound(e_dx, c_out * k * k, dx64), (g_dx, e_dx)
    assert g_dxa < bound(e_dx, c_out * k * k, dx64) + 1e-6, (g_dxa, e_dx)
    assert g_dw < bound(e_dw, n * h * w, dw64), (g_dw, e_dw)
    assert max(g_y, g_dx) < 5e-5
    # exact products AND one rounding of the running sum per 32-channel tap (the nine limb products of a step are summed apart, from
    # zero): as close to fp64 as PyTorch's own fp32 convolution (measured 0.1 ... 1.4x its error, the larger ratios on the longest
    # sums -- one serial chain per output here, blocked partial sums there; 2.8 ... 4.6x when the nine products were accumulated
    # straight into the running sum)
    assert g_y < 2.5 * e_y + 2e-7 and g_dx < 2.5 * e_dx + 2e-7, (g_y, e_y, g_dx, e_dx)


@pytest.mark.parametrize("shape", [(48, 64, 10, 20, 3), (56, 100, 7, 33, 3), (100, 170, 9, 12, 3), (49, 81, 16, 16, 3), (136, 96, 5, 44, 1),
                                   (100, 337, 8, 26, 3)])
def test_three_limb_conv_odd_shapes_against_fp64(shape, l3):
    """Channel counts around every tile / octet / chunk edge (remainder chunks, half-empty wave groups, short last blocks),
    odd planes, several images per box; at the smallest batch that has a persistent tiling (>= 128 work items) in both directions."""
    c_in, c_out, h, w, k = shape
    n = None
    for cand in (8, 16, 32, 64, 128):
        d = F._desc(cand, c_in, c_out, h, w, k, k, l3)
        if F.conv_l16_supported(d, 0) and F.conv_l16_supported(d, 1):
            n = cand
            break
    assert n is not None, shape
    torch.manual_seed(sum(shape))
    pad = k // 2
    x = torch.randn(n, c_in, h, w)
    wt = torch.randn(c_out, c_in, k, k) / (c_in * k * k) ** 0.5
    b = torch.randn(c_out)
    gy = torch.randn(n, c_out, h, w)
    y64 = TF.conv2d(x.double(), wt.double(), b.double(), padding=pad)
    dx64 = torch.nn.grad.conv2d_input(x.shape, wt.double(), gy.double(), padding=pad)
    xd, wd, bd, gd = x.to(DEV), wt.to(DEV), b.to(DEV), gy.to(DEV)
    y = F.conv_l16(F.l16_pack(xd), wd, bd).cpu()
    dx = F.conv_l16(F.l16_pack(gd), wd, None, dgrad=True).cpu()
    assert float((y.double() - y64).abs().max()) < 2e-5
    assert float((dx.double() - dx64).abs().max()) < 2e-5
    if F.conv_l16_wgrad_supported(d):
        dw64 = torch.nn.grad.conv2d_weight(x.double(), wt.shape, gy.double(), padding=pad)
        dw = F.conv_l16_wgrad(F.l16_pack(xd), F.l16_pack(gd), wt.shape).cpu()
        assert float((dw.double() - dw64).abs().max()) < 3e-6 * float(dw64.abs().max()) + 1e-5


def _BN(c, gen):
    bn = torch.nn.BatchNorm2d(c).to(DEV)
    with torch.no_grad():
        bn.weight.copy_(1.0 + 0.3 * torch.randn(c, device=DEV, generator=gen))
        bn.bias.copy_(0.2 * torch.randn(c, device=DEV, generator=gen))
    return bn.train()


@pytest.mark.parametrize("shape", [(4, 100, 16, 43), (3, 37, 9, 20), (2, 150, 32, 107), (5, 24, 1, 300), (128, 48, 3, 5)])
def test_producers_write_the_limbs_of_what_they_write_as_fp32(shape, l3):
    """BatchNorm + PReLU forward, its backward and the backward fused with the max-pool un-pooling: the three-limb output,
    recombined, equals the fp32 output of the same call bit for bit (bf16 limbs are exact; nothing is scaled or bounded) -- with
    scaled fp16 limbs, bit for bit for every element within 2^-16 of the declared maximum and to 2^-39 of that maximum below."""
    n, c, h, w = shape

    def same(t, ref):
        assert t.limbs == FMT[l3]
        back = F.l16_unpack(t)
        if l3 == 9:
            return torch.equal(back, ref)
        top = float(t.amax.max())                              # the declared bound (>= the largest element)
        assert top >= float(ref.abs().max()) and top <= 4.0 * float(ref.abs().max()) + 1e-30
        big = ref.abs() >= 2.0 ** -15 * top
        return torch.equal(back[big], ref[big]) and float((back - ref).abs().max()) <= 2.0 ** -39 * top

    gen = torch.Generator(device=DEV).manual_seed(n + c + h + w)
    x = torch.randn(n, c, h, w, device=DEV, generator=gen)
    bn = _BN(c, gen)
    alpha = 0.25 + 0.1 * torch.rand(c, device=DEV, generator=gen)
    st = F.bn_prepare(x, bn, True)
    y, t = F.bn_act_forward(x, st, alpha, l16=True, want_f32=True)
    assert t is not None and same(t, y)
    y_plain = F.bn_act_forward(x, st, alpha)
    assert torch.equal(y_plain, y)
    # backward
    dy = torch.randn(n, c, h, w, device=DEV, generator=gen) * torch.exp2(torch.randint(-30, 1, (n, 1, 1, 1), device=DEV, generator=gen).float())
    res = F.bn_act_backward(dy, x, st, bn, alpha, want_dres=False, want_chan_sum=True, with_amax=True, l16=True, want_f32=True)
    dx, t2 = res[0], res[-1]
    assert isinstance(t2, F.L16) and same(t2, dx)
    ref = F.bn_act_backward(dy, x, st, bn, alpha, want_dres=False, want_chan_sum=True)
    # (small tensors: the call without limb output is the one-workgroup-per-channel kernel -- the same sums in the same order, the
    # products contracted into other fused multiply-adds: an ulp; tests/test_bn_fused_gpu.py)
    unit = F._lib.load().fsc_bn_train_act_fwd_supported(n, c, h * w) != 0
    for a, b in zip((ref[0],) + tuple(ref[2:5]), (dx,) + tuple(res[2:5])):
        assert torch.equal(a, b) or (unit and float((a - b).abs().max()) <= 2e-6 * float(b.abs().max()))
    # backward + un-pooling (2 x 2 windows; 1 x 2 on single rows)
    ph = 2 if h >= 2 else 1
    c_shape = (n, c, h * ph if ph == 2 else h, 2 * w + 1)
    cfull = torch.randn(c_shape, device=DEV, generator=gen)
    pooled, idx = F.maxpool_forward(cfull, ph)
    assert pooled.shape[2:] == (h if ph == 2 else h, w)
    stp = F.bn_prepare(pooled, bn, True)
    out = F.bn_act_backward_unpool(dy, pooled, stp, bn, alpha, idx, c_shape, ph, l16=True, want_f32=True)
    dc, t3 = out[0], out[-1]
    assert isinstance(t3, F.L16) and same(t3, dc)
    ref = F.bn_act_backward_unpool(dy, pooled, stp, bn, alpha, idx, c_shape, ph)
    # (the un-pooling kernel without limb output is a different kernel with a different thread layout; same arithmetic)
    assert float((ref[0] - dc).abs().max()) <= 1e-6 * float(dc.abs().max())


@pytest.mark.parametrize("shape", [(64, 100, 150, 16, 48), (8, 150, 225, 32, 107), (128, 64, 100, 8, 24)])
def test_three_limb_conv_fused_with_maxpool_and_statistics(shape, l3):
    """fsc_conv_l16_pool_fwd(_stats) on three-limb operands: pooled values and window indices equal conv + fsc_maxpool_fwd bit
    for bit; the statistics of the epilogue (plain and pooled) finalise to the mean / invstd of the separate pass."""
    n, c_in, c_out, h, w = shape
    gen = torch.Generator(device=DEV).manual_seed(sum(shape))
    x = torch.randn(n, c_in, h, w, device=DEV, generator=gen)
    wt = torch.randn(c_out, c_in, 3, 3, device=DEV, generator=gen) / (c_in * 9) ** 0.5
    b = torch.randn(c_out, device=DEV, generator=gen)
    t = F.l16_pack(x)
    full = F.conv_l16(t, wt, b)
    pooled_ref, idx_ref = F.maxpool_forward(full, 2)
    got = F.conv_l16_pool(t, wt, b)
    assert got is not None, shape
    pooled, idx, cshape = got
    assert tuple(cshape) == tuple(full.shape)
    assert torch.equal(pooled, pooled_ref) and torch.equal(idx, idx_ref)
    # epilogue statistics against the separate pass
    bn = _BN(c_out, gen)
    st_ref = F.bn_prepare(full, bn, True)
    full2 = F.conv_l16(t, wt, b, stats_bn=(bn, True))
    assert torch.equal(full2, full)
    st = F.bn_prepare(full2, bn, True)
    assert float(((st.mean - st_ref.mean).abs() * st_ref.invstd).max()) < 1e-5
    assert float((st.invstd / st_ref.invstd - 1).abs().max()) < 1e-5
    stp_ref = F.bn_prepare(pooled_ref, bn, True)
    got2 = F.conv_l16_pool(t, wt, b, stats_bn=(bn, True))
    stp = F.bn_prepare(got2[0], bn, True)
    assert torch.equal(got2[0], pooled_ref)
    assert float(((stp.mean - stp_ref.mean).abs() * stp_ref.invstd).max()) < 1e-5
    assert float((stp.invstd / stp_ref.invstd - 1).abs().max()) < 1e-5


def test_what_the_scaled_limbs_give_up_operand_range():
    """The honest difference between the two three-limb arithmetics.  Images of one batch scaled by 2^0 ... 2^-24: with bf16 limbs
    (no scale) every image's output is as accurate as the first's; with scaled fp16 limbs an image keeps full fp32 accuracy while its
    elements stay within ~2^-16 of the TENSOR's maximum and degrades linearly below (absolute error <= 2^-39 of the tensor maximum per
    operand element) -- the per-tensor scale of the shipped two-limb mode, 2^11 further down."""
    n, c_in, c_out, h, w = 32, 64, 64, 16, 64
    torch.manual_seed(11)
    x = torch.randn(n, c_in, h, w)
    shifts = torch.arange(n).clamp(max=24).float()
    x = x * torch.exp2(-shifts).view(n, 1, 1, 1)
    wt = torch.randn(c_out, c_in, 3, 3) / (c_in * 9) ** 0.5
    y64 = TF.conv2d(x.double(), wt.double(), None, padding=1)
    rms64 = y64.pow(2).mean(dim=(1, 2, 3)).sqrt()
    mode0 = F.get_conv_arith()
    rel = {}
    try:
        for arith in (9, 10):
            F.set_conv_arith(arith)
            assert F.conv_l16_supported(F._desc(n, c_in, c_out, h, w, 3, 3, arith), 0)
            y = F.conv_l16(F.l16_pack(x.to(DEV)), wt.to(DEV), None).cpu().double()
            rel[arith] = ((y - y64).pow(2).mean(dim=(1, 2, 3)).sqrt() / rms64)
    finally:
        F.set_conv_arith(mode0)
    _report("operand range: per-image relative rms error of a 64 -> 64 3x3 convolution, image i scaled by 2^-i: "
            + "; ".join("2^-%d: bf16x9 %.1e f16x6 %.1e" % (i, float(rel[9][i]), float(rel[10][i])) for i in (0, 8, 12, 16, 20, 24)))
    assert float(rel[9].max()) < 3e-7                                     # exact limbs: no dependence on the range
    assert float(rel[10][:13].max()) < 3e-7                               # within 2^-12 of the maximum: indistinguishable
    for i in range(13, n):
        k = int(shifts[i])
        # |x| of image i ~ 2^-k |x|max / 4: operand error <= 2^-39 |x|max each => relative <= 2^-37 2^k; measured ~10x below
        assert float(rel[10][i]) < max(3e-7, 2.0 ** (k - 37)), (i, float(rel[10][i]))


@pytest.fixture(params=[3, 9, 10], ids=["f16x3", "bf16x9", "f16x6"])
def l16_any(request):
    """Every arithmetic with an L16 route (two scaled fp16 limbs, three bf16 limbs, three scaled fp16 limbs)."""
    mode0 = F.get_conv_arith()
    F.set_conv_arith(request.param)
    yield request.param
    F.set_conv_arith(mode0)


@pytest.mark.parametrize("shape", [(64, 100, 100, 16, 53, 3), (16, 150, 150, 32, 107, 1), (128, 337, 337, 8, 26, 3), (128, 96, 170, 9, 20, 3)])
def test_conv_with_the_batchnorm_and_prelu_in_its_epilogue_equals_the_two_pass_route(shape, l16_any):
    """fsc_conv_l16_fwd_act (inference): the limbs the epilogue writes are, bit for bit, those of convolution -> eval-mode
    BatchNorm + PReLU pass -> limb split with the same declared maximum; pad channels are zero; the largest value written is
    reported exactly.  In all three L16 arithmetics (conv_l16.hip and conv_l3.hip carry the epilogue)."""
    l3 = l16_any
    n, c_in, c_out, h, w, k = shape
    gen = torch.Generator(device=DEV).manual_seed(sum(shape))
    x = torch.randn(n, c_in, h, w, device=DEV, generator=gen)
    wt = torch.randn(c_out, c_in, k, k, device=DEV, generator=gen) / (c_in * k * k) ** 0.5
    b = torch.randn(c_out, device=DEV, generator=gen)
    bn = _BN(c_out, gen).eval()
    with torch.no_grad():
        bn.running_mean.normal_(0, 0.3, generator=gen)
        bn.running_var.uniform_(0.5, 1.5, generator=gen)
    alpha = 0.25 + 0.1 * torch.rand(c_out, device=DEV, generator=gen)
    with torch.no_grad():
        assert F.conv_l16_act_supported(x.shape, wt)
        t = F.l16_pack(x)
        r = F.conv_l16(t, wt, b)
        st = F.bn_prepare(r, bn, False)
        y = F.bn_act_forward(r, st, alpha)
        decl = F.amax(y) * 2.0 if l3 != 9 else None
        want = F.l16_pack(y, x_amax=decl)
        seen = torch.zeros(1, device=DEV)
        got = F.conv_l16_act(t, wt, b, st.scale, st.shift, alpha, decl, seen)
    assert got.limbs == want.limbs and torch.equal(got.data, want.data)
    assert float(seen) == float(y.abs().max())
    # without the affine and without the activation: the plain convolution as limbs
    with torch.no_grad():
        decl2 = F.amax(r) if l3 != 9 else None
        got2 = F.conv_l16_act(t, wt, b, None, None, None, decl2, None)
        assert torch.equal(got2.data, F.l16_pack(r, x_amax=decl2).data)


@pytest.mark.parametrize("shape", [(64, 100, 150, 16, 48), (8, 150, 225, 32, 107), (128, 64, 100, 8, 24), (16, 100, 150, 31, 107), (16, 100, 150, 33, 105)])
def test_conv_pool_batchnorm_prelu_in_one_launch_equals_the_three_pass_route(shape, l3):
    """fsc_conv_l16_pool_fwd_act (inference, the entry convolution of a block: reference classifiers.py:526-534): conv 3x3 ->
    MaxPool2d(2) -> eval-mode BatchNorm -> PReLU.  The fp32 result and the limbs are, bit for bit, those of fsc_conv_l16_pool_fwd
    -> fsc_bn_act_fwd -> limb split with the same declared maximum (odd H / W: floor-mode pooling); pad channels zero; the
    largest value written is reported exactly; the fp32 output is optional."""
    n, c_in, c_out, h, w = shape
    gen = torch.Generator(device=DEV).manual_seed(sum(shape))
    x = torch.randn(n, c_in, h, w, device=DEV, generator=gen)
    wt = torch.randn(c_out, c_in, 3, 3, device=DEV, generator=gen) / (c_in * 9) ** 0.5
    b = torch.randn(c_out, device=DEV, generator=gen)
    bn = _BN(c_out, gen).eval()
    with torch.no_grad():
        bn.running_mean.normal_(0, 0.3, generator=gen)
        bn.running_var.uniform_(0.5, 1.5, generator=gen)
    alpha = 0.25 + 0.1 * torch.rand(c_out, device=DEV, generator=gen)
    fold0, F.EVAL_POOL_FOLD = F.EVAL_POOL_FOLD, True              # (off by default: functional.EVAL_POOL_FOLD says why)
    F._L16_OK.clear()
    try:
        _pool_act_case(shape, l3, x, wt, b, bn, alpha)
    finally:
        F.EVAL_POOL_FOLD = fold0
        F._L16_OK.clear()


def _pool_act_case(shape, l3, x, wt, b, bn, alpha):
    n, c_in, c_out, h, w = shape
    with torch.no_grad():
        assert F.conv_l16_pool_act_supported(x.shape, wt), shape
        t = F.l16_pack(x)
        pooled = F.conv_l16_pool(t, wt, b)
        assert pooled is not None
        st = F.bn_prepare(pooled[0], bn, False)
        y = F.bn_act_forward(pooled[0], st, alpha)
        decl = F.amax(y) * 2.0 if l3 != 9 else None
        want = F.l16_pack(y, x_amax=decl)
        seen = torch.zeros(1, device=DEV)
        out, got = F.conv_l16_pool_act(t, wt, b, st.scale, st.shift, alpha, decl, seen)
        assert tuple(out.shape) == (n, c_out, h // 2, w // 2) and torch.equal(out, y)
        assert got.limbs == want.limbs and torch.equal(got.data, want.data)
        assert float(seen) == float(y.abs().max())
        none, got2 = F.conv_l16_pool_act(t, wt, b, st.scale, st.shift, alpha, decl, None, want_f32=False)
        assert none is None and torch.equal(got2.data, want.data)


@pytest.mark.parametrize("arith", ["bf16x9", "f16x6", "f16x3"])
def test_inference_with_folded_residual_units_against_the_two_pass_route(arith):
    """Eval-mode forward of the cfg-2 network (batch 32 x 10 s: the first two blocks have three-limb tilings) with conv -> BatchNorm -> PReLU of the residual units as one launch:
    bf16 limbs -- the same logits (to the 1e-7 run-to-run noise of the head's reductions; the limbs themselves are bit-identical: the
    test above), no scope needed; scaled fp16 limbs -- the first batch calibrates on the two-pass route, the
    second runs folded inside a scope that is ok(), logits within 1e-5; a calibration that is too small is caught by ok(), dropped,
    and the next forward is two-pass again."""
    from freesound_classification_amd.networks.classifiers import TwoDimensionalCNNClassificationModel
    from test_cfg2_gpu import cfg2_experiment
    mode0 = F.get_conv_arith()
    F.set_conv_arith(arith)
    fold0 = F.EVAL_ACT_FOLD

    def same(a, b):
        return float((a - b).abs().max()) < 1e-6
    try:
        torch.manual_seed(3)
        m = TwoDimensionalCNNClassificationModel(cfg2_experiment(), device="cuda:0").eval()
        for mod in m.modules():
            if isinstance(mod, torch.nn.modules.batchnorm._BatchNorm):
                mod.running_mean.normal_(0, 0.1)
                mod.running_var.uniform_(0.5, 1.5)
        x = 0.1 * torch.randn(32, 441000, 1, device=DEV)
        with torch.no_grad():
            F.EVAL_ACT_FOLD = False
            ref = m(x)["class_logits"].clone()
            F.EVAL_ACT_FOLD = True
            F._ACT_CAL.clear()
            calls = []
            orig = F.conv_l16_act
            F.conv_l16_act = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
            try:
                if arith == "bf16x9":
                    out = m(x)["class_logits"]
                    assert len(calls) >= 4 and same(out, ref)
                else:
                    first = m(x)["class_logits"]                 # calibrates (no scope: two-pass)
                    assert not calls and same(first, ref) and len(F._ACT_CAL) >= 4
                    scope = F.act_fold_scope()
                    with scope:
                        out = m(x)["class_logits"]
                    assert len(calls) >= 4 and scope.ok()
                    assert float((out - ref).abs().max()) < 1e-5 * max(1.0, float(ref.abs().max()))
                    # a calibration that is too small: caught, dropped, two-pass again
                    key = next(iter(F._ACT_CAL))
                    ent = F._ACT_CAL[key]
                    F._ACT_CAL[key] = (ent[0] * 2.0 ** -8, ent[1] * 2.0 ** -8) + tuple(ent[2:])
                    scope = F.act_fold_scope()
                    with scope:
                        m(x)
                    assert not scope.ok() and key not in F._ACT_CAL
                    del calls[:]
                    again = m(x)["class_logits"]
                    assert not calls and same(again, ref)
            finally:
                F.conv_l16_act = orig
    finally:
        F.EVAL_ACT_FOLD = fold0
        F.set_conv_arith(mode0)
        F._ACT_CAL.clear()


def test_fold_models_on_several_streams_give_the_single_stream_ensemble():
    """predict_2d_cnn.ensemble_batch runs the fold models of a batch on FOLD_STREAMS streams (the few-item kernels of one model's late
    blocks leave most of the chip to the other streams): same probabilities as on one stream, the folded route's scope ok() on both."""
    import predict_2d_cnn as drv
    from freesound_classification_amd.networks.classifiers import TwoDimensionalCNNClassificationModel
    from test_cfg2_gpu import cfg2_experiment
    mode0 = F.get_conv_arith()
    streams0 = drv.FOLD_STREAMS
    F.set_conv_arith("f16x6")
    try:
        models = []
        for i in range(3):
            torch.manual_seed(10 + i)
            m = TwoDimensionalCNNClassificationModel(cfg2_experiment(), device="cuda:0").eval()
            for mod in m.modules():
                if isinstance(mod, torch.nn.modules.batchnorm._BatchNorm):
                    mod.running_mean.normal_(0, 0.1)
                    mod.running_var.uniform_(0.5, 1.5)
            models.append(m)
        x = 0.1 * torch.randn(32, 441000, 1, device=DEV)
        F._ACT_CAL.clear()
        drv.FOLD_STREAMS = 1
        ref = drv.ensemble_batch(models, x).clone()               # (calibrates)
        for n in (1, 2, 3):
            drv.FOLD_STREAMS = n
            for _ in range(2):
                scope = F.act_fold_scope()
                out = drv.ensemble_batch(models, x, scope).clone()
                assert scope.ok() and len(scope.keys) >= 12
                assert float((out - ref).abs().max()) < 1e-6
    finally:
        drv.FOLD_STREAMS = streams0
        F.set_conv_arith(mode0)
        F._ACT_CAL.clear()
