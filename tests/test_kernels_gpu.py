"""Kernel-level parity: each HIP entry point (called through the C ABI via the Python host
wrappers) against a plain PyTorch fp32 reference of the same op on the CPU, and against the
golden vectors from the imported reference.  Tolerances are absolute."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

if not torch.cuda.is_available():
    pytest.skip("needs an MI355X", allow_module_level=True)

import torch.nn.functional as TF  # noqa: E402

from freesound_classification_amd import functional as F  # noqa: E402
from freesound_classification_amd.ops import utils as putils  # noqa: E402
from oracle import ref_torch as oref  # noqa: E402

DEV = torch.device("cuda:0")


def dev(x):
    return torch.as_tensor(x).to(DEV)


def maxdiff(a, b):
    return float((a.detach().cpu().double() - b.detach().cpu().double()).abs().max())


# ------------------------------------------------------------------------------ front-end
@pytest.mark.parametrize("desc", ["mel_1024_512_64", "mel_2048_1024_128"])
def test_frontend_logmel_golden(golden, desc):
    g = golden("g1_frontend.npz")
    fb = golden("g2_filterbanks.npz")[desc]
    _, (n_fft, hop, n_mel) = putils.parse_features(desc)
    wav = dev(g[desc + ".wav"])
    bands = F.MelBands(fb, DEV)
    out = F.frontend_logmel(wav, n_fft, hop, bands, freq_channel=True)
    assert out.shape == (wav.shape[0], 2, n_mel, 1 + wav.shape[1] // hop)
    assert maxdiff(out[:, 0], torch.from_numpy(g[desc + ".logmel"])) < 1e-3
    ramp = torch.linspace(-1, 1, n_mel).view(1, n_mel, 1).expand_as(out[:, 1].cpu())
    assert maxdiff(out[:, 1], ramp) < 1e-6
    mag = putils.compute_torch_stft(wav, desc)
    assert maxdiff(mag, torch.from_numpy(g[desc + ".mag"])) < 1e-4


def test_frontend_stft_golden(golden):
    g = golden("g1_frontend.npz")
    desc = "stft_256_128"
    wav = dev(g[desc + ".wav"])
    out = F.frontend_stft(wav, 256, 128, apply_log=True)
    assert maxdiff(out, torch.from_numpy(g[desc + ".logmag"])) < 1e-3
    mag = F.frontend_stft(wav, 256, 128, apply_log=False)
    assert maxdiff(mag, torch.from_numpy(g[desc + ".mag"])) < 1e-4


@pytest.mark.parametrize("n_fft,hop,t", [(64, 16, 1000), (128, 64, 777), (512, 128, 4099), (4096, 1024, 9000),
                                          (2048, 1024, 441000)])
def test_frontend_stft_vs_oracle(n_fft, hop, t):
    torch.manual_seed(n_fft + t)
    n = 3 if t < 100000 else 2
    wav = 0.1 * torch.randn(n, t)
    desc = "stft_%d_%d" % (n_fft, hop)
    ref = oref.stft_magnitude(wav, desc)
    got = F.frontend_stft(wav.to(DEV), n_fft, hop, apply_log=False)
    assert got.shape == ref.shape
    assert maxdiff(got, ref) < 2e-4 * max(1.0, float(ref.max()))


# ------------------------------------------------------------------------------ convolution
CONV_CASES = [
    # n, cin, cout, h, w, kh, kw
    (2, 2, 20, 13, 21, 3, 3),          # stem: direct kernels (c_in <= 4)
    (3, 2, 100, 16, 43, 3, 3),
    (2, 1, 24, 9, 12, 3, 3),
    (2, 4, 33, 8, 10, 3, 3),
    (1, 3, 16, 5, 9, 3, 3),
    (3, 12, 12, 9, 7, 3, 3),
    (2, 20, 30, 16, 37, 3, 3),
    (5, 33, 17, 4, 13, 3, 3),
    (7, 40, 130, 2, 6, 3, 3),
    (2, 100, 100, 8, 43, 3, 3),
    (2, 12, 12, 9, 7, 1, 1),
    (3, 100, 100, 6, 11, 1, 1),
    (9, 37, 150, 2, 3, 1, 1),
    (2, 129, 24, 1, 251, 1, 3),
    (3, 24, 36, 1, 62, 1, 3),
    (3, 24, 24, 1, 31, 1, 1),
]


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_fwd_dgrad_wgrad(case):
    n, cin, cout, h, w, kh, kw = case
    torch.manual_seed(sum(case))
    x = torch.randn(n, cin, h, w)
    wt = torch.randn(cout, cin, kh, kw) / (cin * kh * kw) ** 0.5
    b = torch.randn(cout)
    # the reference in fp64: the CPU's own fp32 convolution is off by up to 1e-4 on these shapes, differently on every host
    # (thread count -> summation order), which made the weight-gradient bound below a coin toss on some boxes
    xr = x.double().requires_grad_()
    wr = wt.double().requires_grad_()
    y = TF.conv2d(xr, wr, b.double(), padding=(kh // 2, kw // 2))
    gy = torch.randn(y.shape)
    y.backward(gy.double())
    got = F.conv_forward(x.to(DEV), wt.to(DEV), b.to(DEV))
    assert maxdiff(got, y) < 2e-5 * (cin * kh * kw) ** 0.5 + 1e-5
    dx = F.conv_dgrad(gy.to(DEV), wt.to(DEV), x.shape)
    assert maxdiff(dx, xr.grad) < 2e-5 * (cout * kh * kw) ** 0.5 + 1e-5
    base = torch.randn_like(x).to(DEV)
    acc = F.conv_dgrad(gy.to(DEV), wt.to(DEV), x.shape, accumulate_into=base.clone())
    assert maxdiff(acc - base, xr.grad) < 1e-4
    dw = F.conv_wgrad(x.to(DEV), gy.to(DEV), wt.shape)
    scale = (n * h * w) ** 0.5
    assert maxdiff(dw, wr.grad) < 3e-6 * scale + 1e-5


# split-bf16 kernel (conv_fwd_x3_kernel): full 32-channel chunks, every remainder-octet count, a 25..31
# channel remainder, several channel blocks, split-K, the k3 1-d form, boxes spanning several images
X3_CASES = [
    (2, 100, 100, 16, 43, 3, 3),
    (3, 150, 150, 18, 27, 3, 3),
    (2, 48, 64, 17, 33, 3, 3),
    (2, 60, 225, 16, 30, 3, 3),
    (4, 64, 48, 1, 256, 1, 3),
    (3, 41, 80, 1, 512, 1, 3),
    (16, 337, 100, 4, 13, 3, 3),
    (2, 32, 48, 16, 16, 3, 3),
    (1, 64, 64, 8, 16, 3, 3),          # 128-pixel tile (one pixel tile per wave)
    (2, 100, 100, 16, 43, 1, 1),       # 1x1: 64-channel chunks, remainder 36 channels
    (2, 200, 150, 9, 30, 1, 1),        # 1x1: remainder of one octet
    (4, 64, 48, 1, 256, 1, 1),         # 1-d k1
]


X3_ALL_DIRECTIONS = {X3_CASES[0], X3_CASES[1], X3_CASES[3], X3_CASES[4]}


@pytest.fixture
def restore_conv_arith():
    mode = F.get_conv_arith()
    yield
    F.set_conv_arith(mode)


@pytest.mark.parametrize("mode", [3, 6, 9])
@pytest.mark.parametrize("case", X3_CASES)
def test_conv_split_bf16_is_fp32_accurate(case, mode, restore_conv_arith):
    """The split-limb kernels (3: two scaled fp16 limbs, 6 / 9: three bf16 limbs) must be as accurate as fp32
    arithmetic: their error against an fp64 convolution is bounded by a small multiple of the error PyTorch's
    own fp32 convolution makes."""
    n, cin, cout, h, w, kh, kw = case
    torch.manual_seed(sum(case))
    x = torch.randn(n, cin, h, w)
    wt = torch.randn(cout, cin, kh, kw) / (cin * kh * kw) ** 0.5
    b = torch.randn(cout)
    gy = torch.randn(n, cout, h, w)
    pad = (kh // 2, kw // 2)
    y64 = TF.conv2d(x.double(), wt.double(), b.double(), padding=pad)
    dx64 = torch.nn.grad.conv2d_input(x.shape, wt.double(), gy.double(), padding=pad)
    y32 = TF.conv2d(x, wt, b, padding=pad)
    dx32 = torch.nn.grad.conv2d_input(x.shape, wt, gy, padding=pad)
    e_fwd32 = float((y32.double() - y64).abs().max())
    e_dx32 = float((dx32.double() - dx64).abs().max())
    rms_fwd32 = float((y32.double() - y64).pow(2).mean().sqrt())

    F.set_conv_arith(mode)
    d = F._desc(n, cin, cout, h, w, kh, kw)
    assert F.plan_name(d, 0).startswith("conv_fwd_x3_kernel"), F.plan_name(d, 0)
    if case in X3_ALL_DIRECTIONS:        # elsewhere the planner may keep dgrad / wgrad on the native kernels
        assert F.plan_name(d, 1).startswith("conv_fwd_x3_kernel"), F.plan_name(d, 1)
        assert F.plan_name(d, 2).startswith("conv_wgrad_x3_kernel"), F.plan_name(d, 2)
    got = F.conv_forward(x.to(DEV), wt.to(DEV), b.to(DEV)).cpu()
    dx = F.conv_dgrad(gy.to(DEV), wt.to(DEV), x.shape).cpu()
    # measured on MI355X: rms error 1.0-1.7x, max error 1.5-2.3x that of the fp32 FMA chain (the bf16 MFMA
    # adds its 32 products and the accumulator with one rounding of a wider intermediate, not 32 roundings)
    assert float((got.double() - y64).abs().max()) < 4.0 * e_fwd32 + 1e-7
    assert float((got.double() - y64).pow(2).mean().sqrt()) < 2.5 * rms_fwd32 + 1e-8
    # the bf16 MFMA truncates its wide intermediate: a systematic offset of -0.2 .. -0.4 ulp (measured -2e-8
    # .. -4.5e-8 at unit scale), an order of magnitude below the rms rounding error
    assert abs(float((got.double() - y64).mean())) < 1e-7
    # (bf16 limbs accumulate twice as many MFMA results per output as fp16 limbs; on an unsplit K = 900 chain
    #  their truncation offset reaches 4.9x the fp32 chain's max error: measured on case (16, 337, 100, 4, 13))
    assert float((dx.double() - dx64).abs().max()) < (4.0 if mode == 3 else 6.0) * e_dx32 + 1e-7
    base = torch.randn_like(x).to(DEV)
    acc = F.conv_dgrad(gy.to(DEV), wt.to(DEV), x.shape, accumulate_into=base.clone())
    assert maxdiff(acc - base, dx) < 1e-5
    # weight gradient: both operands are split in registers (conv_wgrad_x3_kernel) where the shape suits it
    dw64 = torch.nn.grad.conv2d_weight(x.double(), wt.shape, gy.double(), padding=pad)
    dw32 = torch.nn.grad.conv2d_weight(x, wt.shape, gy, padding=pad)
    e_dw32 = float((dw32.double() - dw64).abs().max())
    dw = F.conv_wgrad(x.to(DEV), gy.to(DEV), wt.shape).cpu()
    assert float((dw.double() - dw64).abs().max()) < 3.0 * e_dw32 + 1e-6

    # and the native fp32 MFMA kernel on the same inputs agrees to fp32 rounding
    F.set_conv_arith(0)
    assert F.plan_name(F._desc(n, cin, cout, h, w, kh, kw), 0).startswith("conv_fwd_kernel")
    ref = F.conv_forward(x.to(DEV), wt.to(DEV), b.to(DEV)).cpu()
    assert maxdiff(got, ref) < 4.0 * e_fwd32 + 1e-6


def test_conv_split_bf16_random_shapes_against_native(restore_conv_arith):
    """Planner sweep: 90 random shapes (3x3, k3, 1x1; channel counts around every tile / chunk boundary, odd
    widths, single-row images, batch sizes that do not fill a box) through fwd, dgrad (plain and accumulating)
    and wgrad in the split arithmetic against the native fp32-MFMA kernels on the same device tensors."""
    rng = np.random.RandomState(7)
    chans = [32, 33, 40, 48, 56, 57, 63, 64, 65, 72, 95, 96, 100, 112, 128, 129, 150, 160, 161, 200, 225, 256, 257, 280, 299]
    seen = set()
    for trial in range(90):
        kind = trial % 3
        kh, kw = [(3, 3), (1, 3), (1, 1)][kind]
        cin, cout = int(rng.choice(chans)), int(rng.choice(chans))
        n = int(rng.randint(1, 6))
        if kh == 3:
            h, w = int(rng.randint(2, 20)), int(rng.randint(5, 48))
        else:
            h, w = 1, int(rng.randint(24, 640))      # wide rows with > 256 channels: split-K slices of a few steps
        torch.manual_seed(trial)
        x = torch.randn(n, cin, h, w, device=DEV)
        wt = torch.randn(cout, cin, kh, kw, device=DEV) / (cin * kh * kw) ** 0.5
        b = torch.randn(cout, device=DEV)
        gy = torch.randn(n, cout, h, w, device=DEV)
        base = torch.randn_like(x)
        res = {}
        for mode in (0, 3, 6):
            F.set_conv_arith(mode)
            d = F._desc(n, cin, cout, h, w, kh, kw)
            seen.update(F.plan_name(d, m).split("<")[0] for m in (0, 1, 2))
            res[mode] = (F.conv_forward(x, wt, b), F.conv_dgrad(gy, wt, x.shape),
                         F.conv_dgrad(gy, wt, x.shape, accumulate_into=base.clone()), F.conv_wgrad(x, gy, wt.shape))
        k_red = cin * kh * kw
        tols = (4e-6 * k_red ** 0.5 + 1e-5, 4e-6 * (cout * kh * kw) ** 0.5 + 1e-5, 4e-6 * (cout * kh * kw) ** 0.5 + 1e-5,
                6e-6 * (n * h * w) ** 0.5 + 2e-5)
        for mode in (3, 6):
            for name, a, bb, tol in zip(("fwd", "dgrad", "dgrad+acc", "wgrad"), res[0], res[mode], tols):
                assert maxdiff(a, bb) < tol, (trial, mode, name, (n, cin, cout, h, w, kh, kw), maxdiff(a, bb), tol)
    assert {"conv_fwd_x3_kernel", "conv_wgrad_x3_kernel", "conv_fwd_kernel", "conv_wgrad_kernel"} <= seen, seen


@pytest.mark.parametrize("case", [(6, 793, 412, 1, 525, 1, 3), (8, 593, 795, 1, 509, 1, 3), (5, 379, 738, 1, 536, 1, 3),
                                  (8, 774, 684, 1, 345, 1, 3)])
def test_conv_split_k3_long_reduction_against_native(case, restore_conv_arith):
    """1-d k3 layers with many channels: K slices of several 3-step chunks, where a chunk's last step prefetches
    its operand from the NEXT chunk's input box only two steps after that box was issued (a missing wait there
    produced O(1) errors on exactly these shapes; found by tools/conv_sweep.py)."""
    n, cin, cout, h, w, kh, kw = case
    torch.manual_seed(sum(case))
    x = torch.randn(n, cin, h, w, device=DEV)
    wt = torch.randn(cout, cin, kh, kw, device=DEV) / (cin * kh * kw) ** 0.5
    b = torch.randn(cout, device=DEV)
    gy = torch.randn(n, cout, h, w, device=DEV)
    res = {}
    for mode in (0, 3, 6):
        F.set_conv_arith(mode)
        res[mode] = [(F.conv_forward(x, wt, b), F.conv_dgrad(gy, wt, x.shape)) for _ in range(3)]
    for mode in (3, 6):
        for rep in range(3):
            assert maxdiff(res[mode][rep][0], res[0][0][0]) < 4e-6 * (cin * 3) ** 0.5 + 1e-5, (mode, rep, "fwd")
            assert maxdiff(res[mode][rep][1], res[0][0][1]) < 4e-6 * (cout * 3) ** 0.5 + 1e-5, (mode, rep, "dgrad")


@pytest.mark.parametrize("mode", [3, 6])
def test_conv_split_bf16_exact_on_bf16_representable_inputs(mode, restore_conv_arith):
    """With inputs that are exactly representable in one limb (small integers) and power-of-two
    friendly sums the split kernels reproduce the integer result exactly."""
    F.set_conv_arith(mode)
    torch.manual_seed(3)
    x = torch.randint(-4, 5, (2, 64, 16, 16)).float()
    wt = torch.randint(-3, 4, (48, 64, 3, 3)).float()
    d = F._desc(2, 64, 48, 16, 16, 3, 3)
    assert F.plan_name(d, 0).startswith("conv_fwd_x3_kernel")
    y = TF.conv2d(x, wt, None, padding=1)
    got = F.conv_forward(x.to(DEV), wt.to(DEV), None).cpu()
    assert torch.equal(got, y)


def _amax_value(buf):
    assert buf.numel() == F.AMAX_FLOATS
    return float(buf.max())


@pytest.mark.parametrize("numel", [1, 3, 4, 5, 1023, 4096, 100003, 3_000_001])
def test_amax(numel):
    """fsc_amax: the maximum over the slot buffer is exactly max |x| (any length, negative extremes)."""
    torch.manual_seed(numel)
    x = torch.randn(numel)
    x[numel // 2] = -7.5 if numel % 2 else 6.25
    assert _amax_value(F.amax(x.to(DEV))) == float(x.abs().max())
    assert _amax_value(F.amax(torch.zeros(numel, device=DEV))) == 0.0


@pytest.mark.parametrize("case", [(2, 100, 100, 16, 43, 3, 3), (3, 64, 48, 1, 256, 1, 3), (2, 100, 100, 16, 43, 1, 1)])
def test_conv_split_fp16_power_of_two_scale_invariance(case, restore_conv_arith):
    """The split-fp16 kernels scale each operand by a power of two taken from its largest magnitude, so multiplying
    an operand by 2^k changes nothing but the exponent: results are BIT-IDENTICAL up to that factor, from gradients
    of 1e-18 to activations of 1e12 -- far outside the fp16 range -- and an all-zero operand gives exact zeros."""
    n, cin, cout, h, w, kh, kw = case
    F.set_conv_arith(3)
    torch.manual_seed(11)
    x = torch.randn(n, cin, h, w, device=DEV)
    wt = torch.randn(cout, cin, kh, kw, device=DEV) / (cin * kh * kw) ** 0.5
    gy = torch.randn(n, cout, h, w, device=DEV)
    d = F._desc(n, cin, cout, h, w, kh, kw)
    assert F.plan_name(d, 0).startswith("conv_fwd_x3_kernel") and F.plan_name(d, 0).endswith(",3>")
    y, dx, dw = F.conv_forward(x, wt, None), F.conv_dgrad(gy, wt, x.shape), F.conv_wgrad(x, gy, wt.shape)
    for k in (-60, -17, 13, 40):
        f = 2.0 ** k
        assert torch.equal(F.conv_forward(x * f, wt, None), y * f), k
        assert torch.equal(F.conv_forward(x, wt * f, None), y * f), k
        assert torch.equal(F.conv_dgrad(gy * f, wt, x.shape), dx * f), k
        assert torch.equal(F.conv_wgrad(x * f, gy, wt.shape), dw * f), k
        assert torch.equal(F.conv_wgrad(x, gy * f, wt.shape), dw * f), k
    # a declared maximum inside the same binade gives the same scale; a larger one costs low-order bits only
    am = F.amax(x)
    assert torch.equal(F.conv_forward(x, wt, None, x_amax=am), y)
    loose = F.conv_forward(x, wt, None, x_amax=am * 64.0)
    assert 0 < maxdiff(loose, y) < 1e-4 or torch.equal(loose, y)
    b = torch.randn(cout, device=DEV)
    z = F.conv_forward(torch.zeros_like(x), wt, b)
    assert torch.equal(z, b.view(1, -1, 1, 1).expand_as(z))
    assert float(F.conv_wgrad(x, torch.zeros_like(gy), wt.shape).abs().max()) == 0.0


def test_conv_split_fp16_wide_dynamic_range(restore_conv_arith):
    """Elements far below the tensor's maximum lose low-limb bits gradually (they fall under fp16's normal range
    after the per-tensor scaling): with 2^-20 .. 1 magnitudes inside one tensor the result stays within a few
    fp32 ulps of the LARGEST terms' contribution, i.e. the fp32 error bound of the same sum."""
    F.set_conv_arith(3)
    torch.manual_seed(5)
    n, cin, cout, h, w = 2, 96, 64, 12, 40
    mag = torch.exp2(-20.0 * torch.rand(n, cin, h, w))
    x = torch.randn(n, cin, h, w) * mag
    wt = torch.randn(cout, cin, 3, 3) / (cin * 9) ** 0.5
    y64 = TF.conv2d(x.double(), wt.double(), None, padding=1)
    y32 = TF.conv2d(x, wt, None, padding=1)
    got = F.conv_forward(x.to(DEV), wt.to(DEV), None).cpu()
    e32 = float((y32.double() - y64).abs().max())
    assert float((got.double() - y64).abs().max()) < 4.0 * e32 + 1e-9


@pytest.mark.parametrize("shape", [(4, 30, 24, 40), (3, 17, 9, 13), (8, 64, 1, 1)])
def test_bn_kernels_report_amax(shape, restore_conv_arith):
    """The BN forward / backward kernels report max |output| for the split-fp16 conv kernels that consume it."""
    F.set_conv_arith(3)
    n, c, h, w = shape
    torch.manual_seed(n * c)
    x = torch.randn(n, c, h, w, device=DEV) * 3.0 + 1.0
    bn = torch.nn.BatchNorm2d(c).to(DEV)
    with torch.no_grad():
        bn.weight.uniform_(0.5, 2.0)
        bn.bias.normal_()
    alpha = torch.rand(c, device=DEV) * 0.5
    st = F.bn_prepare(x, bn, True)
    y, y_max = F.bn_act_forward(x, st, alpha, with_amax=True)
    assert _amax_value(y_max) == float(y.abs().max())
    gy = torch.randn_like(x) * 1e-6
    dx, _, _, _, _, _, dx_max = F.bn_act_backward(gy, x, st, bn, alpha, with_amax=True)
    assert _amax_value(dx_max) == float(dx.abs().max())
    F.set_conv_arith(6)
    assert F.bn_act_forward(x, st, alpha, with_amax=True)[1] is None


@pytest.mark.parametrize("case", [(2, 2, 20, 13, 21), (3, 2, 100, 16, 43), (2, 1, 24, 9, 12), (1, 2, 7, 2, 9)])
def test_stem_conv_fused_with_maxpool(case):
    """fsc_conv_pool_fwd == conv 3x3 followed by MaxPool2d(2): values and (after the same tie rule) window indices."""
    n, cin, cout, h, w = case
    torch.manual_seed(sum(case))
    x = torch.randn(n, cin, h, w)
    wt = torch.randn(cout, cin, 3, 3) / (cin * 9) ** 0.5
    b = torch.randn(cout)
    fused = F.conv_pool_forward(x.to(DEV), wt.to(DEV), b.to(DEV))
    assert fused is not None
    p, pidx, c_shape = fused
    assert c_shape == (n, cout, h, w)
    c_dev = F.conv_forward(x.to(DEV), wt.to(DEV), b.to(DEV))
    p_ref, idx_ref = F.maxpool_forward(c_dev, 2)
    assert torch.equal(p.cpu(), p_ref.cpu())               # same FMA order in both stem kernels: bit-equal
    assert torch.equal(pidx.cpu(), idx_ref.cpu())
    y = TF.max_pool2d(TF.conv2d(x, wt, b, padding=1), 2)
    assert maxdiff(p, y) < 2e-5
    assert F.conv_pool_forward(torch.randn(2, 12, 8, 8).to(DEV), torch.randn(12, 12, 3, 3).to(DEV), None) is None


def test_conv_no_bias_and_identity_transpose_check():
    # asymmetric weights catch a swapped row/column in the MFMA C-write
    x = torch.zeros(1, 16, 1, 16)
    for c in range(16):
        x[0, c, 0, c] = 1.0
    wt = (torch.arange(16 * 16, dtype=torch.float32).reshape(16, 16, 1, 1)) / 100.0
    y = TF.conv2d(x, wt)
    got = F.conv_forward(x.to(DEV), wt.to(DEV), None)
    assert maxdiff(got, y) < 1e-5


# ------------------------------------------------------------------------------ BN + PReLU unit
class _BN:
    pass


@pytest.mark.parametrize("shape", [(4, 6, 9, 7), (3, 5, 32, 40), (16, 37, 1, 1), (8, 3, 1, 50)])
@pytest.mark.parametrize("with_alpha,with_res", [(False, False), (True, False), (True, True)])
def test_bn_act_unit(shape, with_alpha, with_res):
    n, c, h, w = shape
    torch.manual_seed(n * c + h)
    x = (torch.randn(shape) * 2 + 3).requires_grad_()
    res = torch.randn(shape).requires_grad_() if with_res else None
    bn = torch.nn.BatchNorm2d(c)
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5)
        bn.bias.normal_()
    prelu = torch.nn.PReLU(c) if with_alpha else None
    if prelu is not None:
        with torch.no_grad():
            prelu.weight.uniform_(0.1, 0.4)
    bn.train()
    z = bn(x)
    if res is not None:
        z = z + res
    y = prelu(z) if prelu is not None else z
    gy = torch.randn_like(y)
    y.backward(gy)

    dbn = torch.nn.BatchNorm2d(c).to(DEV)
    dbn.load_state_dict({k: v for k, v in torch.nn.BatchNorm2d(c).state_dict().items()})
    with torch.no_grad():
        dbn.weight.copy_(bn.weight)
        dbn.bias.copy_(bn.bias)
    alpha = prelu.weight.detach().to(DEV) if prelu is not None else None
    xd = x.detach().to(DEV)
    rd = res.detach().to(DEV) if res is not None else None
    st = F.bn_prepare(xd, dbn, True)
    yd = F.bn_act_forward(xd, st, alpha, rd)
    assert maxdiff(yd, y) < 2e-5
    assert maxdiff(dbn.running_mean, bn.running_mean) < 1e-5
    assert maxdiff(dbn.running_var, bn.running_var) < 1e-4
    assert int(dbn.num_batches_tracked) == 1
    dx, dres, dg, db, dal, csum = F.bn_act_backward(gy.to(DEV), xd, st, dbn, alpha, rd, want_dres=with_res,
                                                    want_chan_sum=True)
    assert maxdiff(dx, x.grad) < 5e-5
    assert maxdiff(dg, bn.weight.grad) < 2e-4
    assert maxdiff(db, bn.bias.grad) < 2e-4
    assert maxdiff(csum, x.grad.sum(dim=(0, 2, 3))) < 2e-4
    if with_alpha:
        assert maxdiff(dal, prelu.weight.grad) < 2e-4
    if with_res:
        assert maxdiff(dres, res.grad) < 2e-5
    # eval mode uses the running statistics
    bn.eval()
    with torch.no_grad():
        ze = bn(x)
    st_e = F.bn_prepare(xd, dbn, False)
    assert maxdiff(F.bn_act_forward(xd, st_e), ze) < 2e-5


def test_bn_act_bwd_with_global_max_head():
    torch.manual_seed(5)
    x = torch.randn(3, 4, 6, 5, requires_grad=True)
    res = torch.randn(3, 4, 6, 5)
    bn = torch.nn.BatchNorm2d(4)
    prelu = torch.nn.PReLU(4)
    out = prelu(bn(x) + res)
    feat = out.flatten(2).amax(2)
    g_out, g_feat = torch.randn_like(out), torch.randn_like(feat)
    (out * g_out).sum().add((feat * g_feat).sum()).backward()
    dbn = torch.nn.BatchNorm2d(4).to(DEV)
    xd, rd = x.detach().to(DEV), res.to(DEV)
    st = F.bn_prepare(xd, dbn, True)
    od = F.bn_act_forward(xd, st, prelu.weight.detach().to(DEV), rd)
    fd, fidx = F.global_maxpool_forward(od)
    assert maxdiff(fd, feat) < 1e-5
    dx, _, dg, db, dal, _ = F.bn_act_backward(g_out.to(DEV), xd, st, dbn, prelu.weight.detach().to(DEV), rd,
                                              gmax=(g_feat.to(DEV), fidx))
    assert maxdiff(dx, x.grad) < 5e-5
    assert maxdiff(dg, bn.weight.grad) < 2e-4
    # head-only gradient (last block): no dense upstream
    x.grad = None
    out = prelu(bn(x) + res)
    (out.flatten(2).amax(2) * g_feat).sum().backward()
    dx2, *_ = F.bn_act_backward(None, xd, st, dbn, prelu.weight.detach().to(DEV), rd, gmax=(g_feat.to(DEV), fidx))
    assert maxdiff(dx2, x.grad) < 5e-5


# ------------------------------------------------------------------------------ pooling
@pytest.mark.parametrize("shape,ph", [((2, 3, 13, 21), 2), ((2, 5, 8, 8), 2), ((3, 4, 1, 17), 1), ((1, 2, 2, 2), 2),
                                      ((128, 476, 1, 6), 1), ((128, 244, 1, 53), 1), ((7, 5, 1, 431), 1), ((3, 4, 1, 1001), 1)])
def test_maxpool(shape, ph):
    # (single rows of <= 256 windows: the thread-per-window kernel over all planes; longer rows and 2-d planes: a workgroup per plane)
    torch.manual_seed(sum(shape))
    x = torch.randn(shape)
    x[0, 0, 0, :4] = 1.5          # ties: the first maximum must win
    x[-1, -1, 0, 1] = float("nan")   # NaN wins (ATen)
    xr = x.clone().requires_grad_()
    y = TF.max_pool2d(xr, (ph, 2), (ph, 2))
    gy = torch.randn_like(y)
    y.backward(gy)
    yd, idx = F.maxpool_forward(x.to(DEV), ph)
    assert torch.equal(torch.isnan(yd.cpu()), torch.isnan(y))
    assert maxdiff(torch.nan_to_num(yd), torch.nan_to_num(y.detach())) == 0.0
    dx = F.maxpool_backward(gy.to(DEV), idx, x.shape, ph)
    assert maxdiff(torch.nan_to_num(dx), torch.nan_to_num(xr.grad)) == 0.0


@pytest.mark.parametrize("shape", [(3, 5, 7, 9), (2, 130, 2, 6), (4, 3, 1, 1000)])
def test_global_maxpool(shape):
    torch.manual_seed(sum(shape))
    x = torch.randn(shape)
    y, idx = F.global_maxpool_forward(x.to(DEV))
    ref, ridx = x.flatten(2).max(dim=2)
    assert maxdiff(y, ref) == 0.0
    assert (idx.cpu().long() == ridx).all()


# ------------------------------------------------------------------------------ head ops
@pytest.mark.parametrize("m,k,n", [(4, 20, 80), (128, 1977, 80), (16, 300, 300), (3, 7, 5), (128, 1977, 1977), (37, 131, 67),
                                   (128, 64, 33), (65, 33, 129)])
def test_linear(m, k, n):
    # (odd row lengths: the 16-byte loads of the tiles are 4-byte aligned; edges in every direction; the head's own sizes)
    torch.manual_seed(m + k + n)
    x = torch.randn(m, k, requires_grad=True)
    w = (torch.randn(n, k) / k ** 0.5).requires_grad_()
    b = torch.randn(n, requires_grad=True)
    y = TF.linear(x, w, b)
    gy = torch.randn_like(y)
    y.backward(gy)
    xd, wd, bd = (t.detach().to(DEV).requires_grad_() for t in (x, w, b))
    yd = F.linear(xd, wd, bd)
    yd.backward(gy.to(DEV))
    assert maxdiff(yd, y) < 1e-4
    assert maxdiff(xd.grad, x.grad) < 1e-4
    assert maxdiff(wd.grad, w.grad) < 1e-4 * max(1.0, m ** 0.5)
    assert maxdiff(bd.grad, b.grad) < 1e-4


def test_dropout_statistics_and_backward():
    x = torch.ones(512, 400, device=DEV, requires_grad=True)
    y = F.dropout(x, 0.7, True)
    kept = (y != 0).float().mean().item()
    assert abs(kept - 0.3) < 0.01
    assert abs(float(y.max()) - 1 / 0.3) < 1e-4
    y.sum().backward()
    assert torch.equal((x.grad != 0), (y != 0))
    assert F.dropout(x, 0.7, False) is x


@pytest.mark.parametrize("n,c", [(128, 80), (5, 256), (3, 300), (7, 64), (4, 65)])
def test_lsep_over_the_pairs_that_exist(n, c):
    """c <= 256: the kernels that list the classes with partners (multi-hot labels: the positives) against the oracle's pairwise
    table (reference networks/losses.py:47-58) -- no positives, everything positive, soft targets, c > 256 (the masked kernels)."""
    from freesound_classification_amd.networks.losses import lsep_loss
    from oracle import ref_torch as oref
    torch.manual_seed(n * c)
    logits = 2.0 * torch.randn(n, c)
    labels = (torch.rand(n, c) < 0.03).float()
    labels[0] = 0.0                                  # no positive: loss 0, gradient 0
    labels[1] = 1.0                                  # nothing below: the same
    labels[2] = torch.rand(c)                        # soft targets: every class but the smallest has partners
    labels[min(3, n - 1), 5] = 1.0
    ref_x = logits.clone().requires_grad_()
    ref = oref.lsep(ref_x, labels, average=False)
    w = torch.rand(n)
    (ref * w).sum().backward()
    x = logits.to(DEV).requires_grad_()
    val = lsep_loss(x, labels.to(DEV), average=False)
    (val * w.to(DEV)).sum().backward()
    assert maxdiff(val, ref) < 2e-5 * max(1.0, float(ref.abs().max()))
    assert maxdiff(x.grad, ref_x.grad) < 2e-5
    assert float(val[0]) == 0.0 and float(x.grad[0].abs().max()) == 0.0 and float(val[1]) == 0.0


def test_losses_golden(golden):
    from freesound_classification_amd.networks.losses import binary_cross_entropy, lsep_loss
    g = golden("g6_losses.npz")
    t = dev(g["targets"])
    for avg, tag in ((True, "avg"), (False, "per")):
        x = dev(g["logits"]).requires_grad_()
        val = lsep_loss(x, t, average=avg)
        (val if avg else F.mean(val, float(val.numel()))).backward()
        assert maxdiff(val, torch.from_numpy(g["lsep_" + tag])) < 2e-5
        assert maxdiff(x.grad, torch.from_numpy(g["lsep_%s_grad" % tag])) < 2e-5
    x = dev(g["logits"]).requires_grad_()
    val = lsep_loss(x, dev(g["soft_targets"]), average=False)
    F.mean(val, float(val.numel())).backward()
    assert maxdiff(val, torch.from_numpy(g["lsep_soft"])) < 2e-5
    assert maxdiff(x.grad, torch.from_numpy(g["lsep_soft_grad"])) < 2e-5
    x = dev(g["logits"]).requires_grad_()
    val = binary_cross_entropy(x, t)
    val.backward()
    assert maxdiff(val, torch.from_numpy(g["bce"])) < 1e-5
    assert maxdiff(x.grad, torch.from_numpy(g["bce_grad"])) < 1e-6
    assert maxdiff(F.sigmoid(x), torch.sigmoid(torch.from_numpy(g["logits"]))) < 1e-6


def test_mixup_batch_bit_exact(golden):
    g = golden("g7_mixup.npz")
    import random
    for case in ("long_first", "short_first", "equal"):
        a, b = g[case + ".a"], g[case + ".b"]
        np.random.seed(70)
        random.seed(70)
        alpha = np.random.uniform(0.4, 0.6)
        start = 0
        if a.size != b.size:
            start = random.randint(0, max(a.size, b.size) - 1 - min(a.size, b.size))
        ya, yb = g[case + ".ya"][None], g[case + ".yb"][None]
        out, lab = F.mixup_batch(dev(a[None]), dev(b[None]), [a.size], [b.size], [start], [alpha],
                                 dev(ya), dev(yb))
        np.testing.assert_array_equal(out.cpu().numpy()[0], g[case + ".mixed"])
        np.testing.assert_array_equal(lab.cpu().numpy()[0], g[case + ".labels"])


def test_optimizers_golden(golden):
    from freesound_classification_amd.ops.training import OPTIMIZERS
    g = golden("g9_optim.npz")
    w = torch.nn.Parameter(dev(g["sgd.w0"]))
    opt = OPTIMIZERS["momentum"]([w], 0.05, weight_decay=1e-3)
    for s in range(3):
        w.grad = dev(g["sgd.g%d" % s])
        opt.step()
        assert maxdiff(w, torch.from_numpy(g["sgd.w%d" % (s + 1)])) < 1e-6
    w = torch.nn.Parameter(dev(g["adam.w0"]))
    opt = OPTIMIZERS["adam"]([w], 0.003, weight_decay=1e-2)
    for s in range(4):
        w.grad = dev(g["adam.g%d" % s])
        opt.step()
        assert maxdiff(w, torch.from_numpy(g["adam.w%d" % (s + 1)])) < 2e-6
    sd = opt.state_dict()["state"][0]
    assert set(sd) == {"step", "exp_avg", "exp_avg_sq", "max_exp_avg_sq"}


@pytest.mark.parametrize("shape,ph", [((3, 5, 13, 21), 2), ((2, 4, 8, 6), 2), ((3, 6, 1, 17), 1)])
def test_bn_act_bwd_fused_with_unpool(shape, ph):
    """fsc_bn_act_bwd_unpool == max_pool backward of (BN + PReLU backward), incl. odd trailing row/col."""
    n, c, h, w = shape
    torch.manual_seed(h * w + c)
    cfull = torch.randn(shape, requires_grad=True)
    bn = torch.nn.BatchNorm2d(c)
    prelu = torch.nn.PReLU(c)
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5)
        bn.bias.normal_()
        prelu.weight.uniform_(0.1, 0.4)
    p = TF.max_pool2d(cfull, (ph, 2), (ph, 2))
    y = prelu(bn(p))
    gy = torch.randn_like(y)
    y.backward(gy)
    dbn = torch.nn.BatchNorm2d(c).to(DEV)
    with torch.no_grad():
        dbn.weight.copy_(bn.weight)
        dbn.bias.copy_(bn.bias)
    pd, pidx = F.maxpool_forward(cfull.detach().to(DEV), ph)
    st = F.bn_prepare(pd, dbn, True)
    alpha = prelu.weight.detach().to(DEV)
    dc, dg, db, dal, csum, dc_max = F.bn_act_backward_unpool(gy.to(DEV), pd, st, dbn, alpha, pidx, shape, ph)
    if dc_max is not None:          # split-fp16 arithmetic: the kernel reports max |dc|
        assert float(dc_max.max()) == float(dc.abs().max())
    assert maxdiff(dc, cfull.grad) < 5e-5
    assert maxdiff(dg, bn.weight.grad) < 2e-4
    assert maxdiff(db, bn.bias.grad) < 2e-4
    assert maxdiff(dal, prelu.weight.grad) < 2e-4
    assert maxdiff(csum, cfull.grad.sum(dim=(0, 2, 3))) < 2e-4
