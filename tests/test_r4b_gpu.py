"""Round 4, second half: the launch-bound workload's kernels (cfg 3: the 1-d model, /root/reference/networks/classifiers.py:107-217 on
stft_256_* features, ops/utils.py:110-127), each against plain PyTorch or against the library's own one-at-a-time form.

* BatchNorm reduce passes on planes of ANY length side by side (odd lengths: quads behind the alignment peel + two edge lanes per plane),
  including the sizes at which the kernel changes form;
* the per-channel sum of dx in closed form (no atomics): equals the sum of the stored dx within its rounding, on one replica and through
  the two phases of the replicated form;
* parameter-gradient-only backward (phase 1, dx = NULL) equals the full backward's parameter gradients bit for bit;
* the n_fft = 256 front-end (eight lanes per frame) against torch.stft, clip edges (reflect padding), ragged last workgroup, odd hop;
* fsc_conv_pack_weights_multi equals fsc_conv_pack_weights bit for bit, both directions; the fragments packed up front reach the
  convolutions of a training step (a step with them equals a step without).
"""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu

if not torch.cuda.is_available():
    pytest.skip("needs an MI355X", allow_module_level=True)

import torch.nn as nn  # noqa: E402

from freesound_classification_amd import _lib  # noqa: E402
from freesound_classification_amd import functional as F  # noqa: E402
from freesound_classification_amd._lib import ConvDesc, call, ptr, stream_ptr  # noqa: E402

DEV = torch.device("cuda:0")

# plane lengths around every form change of the reduce kernels: 1..3 elements per edge lane, the last side-by-side length (1016 for a
# multiple of four, 1015 otherwise: 253 quads + 2 edge lanes = 255 <= 256), the first one-plane-per-trip lengths, the 1-d model's own
ODD_PLANES = [2, 3, 5, 6, 7, 13, 26, 53, 107, 215, 430, 861, 1015, 1016, 1017, 1019, 1020, 1021, 1024, 1723]


def _bn_case(n, c, hw, seed):
    g = torch.Generator(device="cpu").manual_seed(seed)
    x = (torch.randn(n, c, 1, hw, generator=g) * 1.7 + 0.4).to(DEV)
    dy = torch.randn(n, c, 1, hw, generator=g).to(DEV)
    bn = nn.BatchNorm2d(c).to(DEV)
    prelu = nn.PReLU(c).to(DEV)
    with torch.no_grad():
        bn.weight.copy_(torch.rand(c, generator=g) + 0.5)
        bn.bias.copy_(torch.rand(c, generator=g) - 0.5)
        prelu.weight.copy_(torch.rand(c, generator=g) * 0.3 + 0.1)
    return x, dy, bn, prelu


@pytest.mark.parametrize("hw", ODD_PLANES)
def test_bn_reduce_passes_on_planes_of_any_length(hw):
    """Against the unit's formulas in fp64 (nn.BatchNorm2d + nn.PReLU, classifiers.py:533-534).  NOT against torch's own fp32
    backward on the GPU: for odd batch and plane sizes like (37, 11, 1, 861) this torch build's batch-norm backward returns a wrong
    bias gradient (off by 13 - 20 absolute, tools/dbg_bn_planes.py), which is how the first version of this test failed."""
    n, c = 37, 11                                       # (odd batch: trips past the batch; planes start at every 4-byte offset)
    x, dy, bn, prelu = _bn_case(n, c, hw, hw)
    gamma, beta, alpha = (t.detach().double().view(1, -1, 1, 1) for t in (bn.weight, bn.bias, prelu.weight))
    xd, dyd = x.double(), dy.double()
    mean = xd.mean((0, 2, 3), keepdim=True)
    var = xd.var((0, 2, 3), unbiased=False, keepdim=True)
    invstd = 1.0 / torch.sqrt(var + bn.eps)
    xh = (xd - mean) * invstd
    z = xh * gamma + beta
    y_ref = torch.where(z > 0, z, alpha * z)
    dz = torch.where(z > 0, dyd, alpha * dyd)
    dx_ref = gamma * invstd * (dz - dz.mean((0, 2, 3), keepdim=True) - xh * (dz * xh).mean((0, 2, 3), keepdim=True))
    st = F.bn_prepare(x, bn, True)
    assert (st.mean.double() - mean.flatten()).abs().max().item() < 2e-6
    assert (st.invstd.double() / invstd.flatten() - 1).abs().max().item() < 2e-6
    cnt = n * hw
    rv_ref = 0.9 + 0.1 * var.flatten() * cnt / (cnt - 1)
    assert (bn.running_mean.double() - 0.1 * mean.flatten()).abs().max().item() < 1e-6
    assert (bn.running_var.double() - rv_ref).abs().max().item() < 1e-5
    y = F.bn_act_forward(x, st, prelu.weight)
    assert (y.double() - y_ref).abs().max().item() < 2e-5
    dx, _, dg, db, dal, csum = F.bn_act_backward(dy, x, st, bn, prelu.weight, want_chan_sum=True)
    assert (dx.double() - dx_ref).abs().max().item() < 2e-6 * max(1.0, dx_ref.abs().max().item())
    tol = 1e-6 * cnt ** 0.5 * 8
    assert (db.double() - dz.sum((0, 2, 3))).abs().max().item() < tol
    assert (dg.double() - (dz * xh).sum((0, 2, 3))).abs().max().item() < tol
    assert (dal.double() - (dyd * torch.where(z > 0, torch.zeros_like(z), z)).sum((0, 2, 3))).abs().max().item() < tol
    # the closed form against the sum of the dx this very call stored (both are rounding residue of an analytically zero sum)
    assert (csum.double() - dx.double().sum((0, 2, 3))).abs().max().item() < 1e-6 * cnt ** 0.5 * dx.abs().max().item() + 1e-7 * cnt


def test_channel_sum_of_dx_through_the_two_phases_is_the_replicas_own():
    """Two 'replicas' (halves of a batch) with all-reduced statistics: each half's dx_chan_sum is the sum of ITS dx (the gradient
    all-reduce adds the replicas), and the two add up to the single-replica sum."""
    n, c, hw = 24, 13, 215
    x, dy, bn, prelu = _bn_case(n, c, hw, 5)
    halves = [(x[:9].contiguous(), dy[:9].contiguous()), (x[9:].contiguous(), dy[9:].contiguous())]       # uneven on purpose
    bufs = []

    def sync(buf):
        bufs.append(buf)

    # statistics over the whole batch (what SyncBN computes), then each half backward with the other's sums added in `sync`
    st = F.bn_prepare(x, bn, True)
    pending = []
    for xh, dyh in halves:
        pending.append(F.bn_act_backward(dyh, xh, st, bn, prelu.weight, want_chan_sum=True, want_dx=False))   # phase 1 sums only
    # run the two-phase form by hand: phase 1 on both, add, phase 2 on both
    sums, args = [], []
    for xh, dyh in halves:
        nh = xh.shape[0]
        dx = torch.empty_like(xh)
        out = {k: torch.empty(c, device=DEV) for k in ("dg", "db", "dal", "csum")}
        ws = torch.zeros((_lib.load().fsc_bn_workspace_bytes(c) + 7) // 8, device=DEV, dtype=torch.float64)
        sm = torch.empty(4 * c, device=DEV, dtype=torch.float64)
        a = (ptr(dyh), None, None, ptr(xh), None, ptr(st.mean), ptr(st.invstd), ptr(bn.weight), ptr(bn.bias), ptr(prelu.weight), ptr(dx),
             None, ptr(out["dg"]), ptr(out["db"]), ptr(out["dal"]), ptr(out["csum"]), nh, c, hw, ptr(ws), None)
        call("fsc_bn_act_bwd", *a, ptr(sm), 1, None, stream_ptr())
        sums.append(sm)
        args.append((a, dx, out, ws, sm))
    total = sums[0] + sums[1]
    for a, dx, out, ws, sm in args:
        sm.copy_(total)
        call("fsc_bn_act_bwd", *a, ptr(sm), 2, None, stream_ptr())
    full = F.bn_act_backward(dy, x, st, bn, prelu.weight, want_chan_sum=True)
    dx_cat = torch.cat([args[0][1], args[1][1]])
    assert (dx_cat - full[0]).abs().max().item() < 1e-5 * max(1.0, full[0].abs().max().item())
    for a, dx, out, ws, sm in args:
        own = dx.double().sum((0, 2, 3))
        assert (out["csum"].double() - own).abs().max().item() < 1e-5 * dx.abs().max().item() * (dx.shape[0] * hw) ** 0.5
    assert ((args[0][2]["dg"] + args[1][2]["dg"]) - full[2]).abs().max().item() < 1e-4 * max(1.0, full[2].abs().max().item())
    for p, (a, dx, out, ws, sm) in zip(pending, args):            # parameter-only calls: the same local sums
        assert torch.equal(p[2], out["dg"]) and torch.equal(p[3], out["db"]) and torch.equal(p[4], out["dal"])
        assert p[0] is None


@pytest.mark.parametrize("shape", [(128, 129, 1, 3446), (16, 64, 1, 215), (8, 24, 9, 11)])
def test_parameter_only_backward_equals_the_full_backward(shape):
    n, c, h, w = shape
    g = torch.Generator(device="cpu").manual_seed(3)
    x = torch.randn(n, c, h, w, generator=g).to(DEV)
    dy = torch.randn(n, c, h, w, generator=g).to(DEV)
    bn = nn.BatchNorm2d(c).to(DEV)
    st = F.bn_prepare(x, bn, True)
    full = F.bn_act_backward(dy, x, st, bn)
    only = F.bn_act_backward(dy, x, st, bn, want_dx=False)
    assert only[0] is None
    assert torch.equal(full[2], only[2]) and torch.equal(full[3], only[3])


@pytest.mark.parametrize("t,hop,n", [(441000, 128, 3), (4096 + 37, 128, 2), (1000, 64, 2), (300, 100, 1), (129, 128, 2), (8000, 77, 2)])
@pytest.mark.parametrize("apply_log", [True, False])
def test_frontend_256_against_torch_stft(t, hop, n, apply_log):
    g = torch.Generator(device="cpu").manual_seed(t + hop)
    wav = (0.1 * torch.randn(n, t, generator=g)).to(DEV)
    wav[-1, t // 2:] = 0.0                                       # a zero-padded tail: frames of exactly log(eps)
    out = F.frontend_stft(wav, 256, hop, apply_log)
    # (the reference on the CPU -- torch.stft as the oracle restates ops/utils.py:110-127 -- not rocFFT on the same GPU)
    ref = torch.stft(wav.cpu(), 256, hop_length=hop, window=torch.hann_window(256), center=True, pad_mode="reflect",
                     return_complex=True).abs().to(DEV)
    assert out.shape == ref.shape == (n, 129, 1 + t // hop)
    if apply_log:
        # |log(a + eps) - log(b + eps)| <= |a - b| / eps in the worst case: compare the magnitudes through the inverse map
        got = torch.exp(out) - F.LOG_EPS
        assert (got - ref).abs().max().item() < 2e-5 * max(1.0, ref.max().item())
        if t >= 4096:                                            # the last frame lies wholly inside the zero tail: exactly log(eps)
            assert (out[-1, :, -1] - torch.log(torch.tensor(F.LOG_EPS))).abs().max().item() < 1e-5
    else:
        assert (out - ref).abs().max().item() < 2e-5 * max(1.0, ref.max().item())


def test_frontend_256_frequency_channel():
    wav = 0.1 * torch.randn(2, 5000, device=DEV)
    out = F.frontend_stft(wav, 256, 128, True, freq_channel=True)
    plain = F.frontend_stft(wav, 256, 128, True)
    assert torch.equal(out[:, 0], plain)
    lin = torch.linspace(-1, 1, 129, device=DEV)
    assert (out[:, 1] - lin[None, :, None]).abs().max().item() < 1e-6


CFG3_CONVS = [(128, 129, 64, 3446, 3), (128, 64, 64, 1723, 1), (128, 100, 125, 430, 3), (128, 156, 156, 107, 1), (128, 305, 381, 13, 3),
              (128, 476, 476, 3, 3), (128, 476, 476, 3, 1)]


def test_multi_pack_equals_single_pack_bit_for_bit():
    lib = _lib.load()
    jobs = []
    g = torch.Generator(device="cpu").manual_seed(11)
    for (n, ci, co, l, k) in CFG3_CONVS:
        w = torch.randn(co, ci, 1, k, generator=g).to(DEV)
        for dgrad in (0, 1):
            d = ConvDesc(n, ci, co, 1, l, 1, k, 1)
            assert lib.fsc_conv_pack_weights_multi_supported(C.byref(d), dgrad), (n, ci, co, l, k, dgrad)
            nfl = lib.fsc_conv_packed_floats(C.byref(d), dgrad)
            single = torch.zeros(nfl, device=DEV)
            call("fsc_conv_pack_weights", C.byref(d), ptr(w), dgrad, ptr(single), stream_ptr())
            jobs.append((d, w, dgrad, single, torch.zeros(nfl, device=DEV)))
    jobs = jobs * 4                                              # 56 jobs: more than one launch's table
    jobs = [(d, w, dg, s, torch.zeros_like(s)) for (d, w, dg, s, _) in jobs]
    count = len(jobs)
    descs = (ConvDesc * count)()
    wp, pp, dg = (C.c_void_p * count)(), (C.c_void_p * count)(), (C.c_int * count)()
    for i, (d, w, dgrad, single, multi) in enumerate(jobs):
        descs[i] = d
        wp[i], pp[i], dg[i] = ptr(w), ptr(multi), dgrad
    call("fsc_conv_pack_weights_multi", count, descs, wp, dg, pp, stream_ptr())
    for d, w, dgrad, single, multi in jobs:
        assert torch.equal(single.view(torch.int32), multi.view(torch.int32)), (d.c_in, d.c_out, d.w, d.kw, dgrad)
    # a job that is not a bf16-limb tiling fails the call loudly
    bad = ConvDesc(128, 100, 100, 64, 215, 3, 3, 3)
    assert not lib.fsc_conv_pack_weights_multi_supported(C.byref(bad), 0)


def test_mixup_rows_equals_mixup_batch_on_gathered_partners():
    """The partner-table form (no gathered copy; undrawn rows pass through) against mixup_batch on explicitly gathered rows, bit for bit
    (ops/audio.py:32-52; a row mixed with itself at equal length is (x + x) / 2 = x, what bench.py's cfg-3 loop used before)."""
    import numpy as np
    rng = np.random.RandomState(3)
    n, t, c = 12, 4001, 80
    a = torch.randn(n, t, device=DEV)
    lens = rng.randint(t // 2, t + 1, size=n)
    lens[:4] = t                                                   # some equal-length pairs
    for i in range(n):
        a[i, lens[i]:] = 0.0
    labels = (torch.rand(n, c, device=DEV) < 0.05).float()
    perm = rng.permutation(n)
    take = rng.uniform(size=n) < 0.6
    partner = np.where(take, perm, -1)
    start = [int(rng.randint(0, max(1, abs(int(lens[i]) - int(lens[perm[i]])) + 1))) for i in range(n)]
    alpha = rng.uniform(0.4, 0.6, size=n)
    got, got_l = F.mixup_rows(a, a, partner, lens, lens[perm], start, alpha, labels, labels)
    idx = torch.from_numpy(np.where(take, perm, np.arange(n))).to(DEV)
    len_b = np.where(take, lens[perm], lens)
    want, want_l = F.mixup_batch(a, a[idx].contiguous(), lens, len_b, start, alpha, labels, labels[idx].contiguous())
    assert torch.equal(got, want)
    assert torch.equal(got_l, want_l)
    for i in range(n):
        if not take[i]:
            assert torch.equal(got[i], a[i])


def test_deferred_weight_gradient_reduces_equal_the_immediate_ones():
    """fsc_conv_wgrad_partial + fsc_conv_wgrad_reduce_multi (functional.wgrad_begin / wgrad_flush) against fsc_conv_wgrad: the same
    additions in the same order, bit-identical -- on both weight-gradient kernels of the fp32-input path, more jobs than one launch's
    table, and a deferral that an exception leaves behind does not outlive its block."""
    cases = [("bf16", 128, 64, 64, 1, 1723, 3), ("bf16", 128, 476, 476, 1, 3, 1), ("bf16", 128, 305, 381, 1, 13, 3),
             ("f16x3", 128, 759, 759, 2, 6, 3), ("f16x3", 128, 506, 506, 4, 13, 1)]
    for arith, n, ci, co, h, w, k in cases:
        F.set_conv_arith(arith)
        try:
            g = torch.Generator(device="cpu").manual_seed(ci + w)
            x = torch.randn(n, ci, h, w, generator=g).to(DEV)
            dy = torch.randn(n, co, h, w, generator=g).to(DEV)
            shape = (co, ci, k if h > 1 else 1, k)
            want = F.conv_wgrad(x, dy, shape)
            F.wgrad_begin()
            try:
                got = [F.conv_wgrad(x, dy, shape) for _ in range(18)]       # 18 jobs: two launches
                assert len(F._WGRAD.pending) == 18
                F.wgrad_flush()
            finally:
                F.wgrad_abort()
            assert F._WGRAD.pending is None
            for t in got:
                assert torch.equal(t, want), (arith, n, ci, co, h, w, k)
        finally:
            F.set_conv_arith(None)
