"""The statistics epilogue of the fp32-input ring kernels (plain bf16, 1-d rows: fsc_conv_fwd_stats): the BatchNorm that reads a
Conv1d output (networks/classifiers.py:78-101 on the 1-d model, 147-163) gets its batch statistics from the convolution's
launch; same output bits as fsc_conv_fwd, statistics equal to the separate pass to fp32 summation accuracy."""
import ctypes as C

import pytest
import torch

from freesound_classification_amd import functional as F, _lib

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture
def bf16():
    mode = F.get_conv_arith()
    F.set_conv_arith("bf16")
    yield
    F.set_conv_arith(mode)


def _layout(d):
    out4 = (C.c_int * 4)()
    return tuple(out4) if _lib.load().fsc_conv_fwd_stats_layout(C.byref(d), out4) else None


# cfg 3's blocks 0 - 2 at batch 128 (k3 and k1), a batch that does not fill the workers, a shape whose last tile overhangs the row
CASES = [(128, 64, 64, 1723, 3), (128, 64, 64, 1723, 1), (128, 80, 80, 861, 3), (128, 100, 100, 430, 3), (128, 100, 100, 430, 1), (32, 64, 80, 1500, 3),
         (16, 48, 64, 3001, 3)]


@pytest.mark.parametrize("case", CASES)
def test_statistics_from_the_convolution_launch(case, bf16):
    n, ci, co, length, k = case
    torch.manual_seed(sum(case))
    x = torch.randn(n, ci, 1, length, device=DEV)
    w = torch.randn(co, ci, 1, k, device=DEV) / (ci * k) ** 0.5
    b = torch.randn(co, device=DEV) * 3.0                     # (channel means far from zero: the pivot matters)
    d = F._desc(n, ci, co, 1, length, 1, k)
    assert F.plan_name(d, 0).startswith("conv_fwd_x3_kernel"), F.plan_name(d, 0)
    assert _layout(d) is not None
    ref = F.conv_forward(x, w, b)
    out = []
    for with_stats in (False, True):
        bn = torch.nn.BatchNorm2d(co).to(DEV).train()
        with torch.no_grad():
            bn.running_mean.copy_(b + 0.3)                    # (a pivot near the batch mean, as in steady-state training)
            bn.weight.uniform_(0.5, 1.5)
        F._PRESTATS.clear()
        y = F.conv_forward(x, w, b, stats_bn=(bn, True) if with_stats else None)
        assert bool(F._PRESTATS) == with_stats
        st = F.bn_prepare(y, bn, True)
        assert not F._PRESTATS
        out.append((y, st.mean, st.invstd, st.scale, st.shift, st.minmax, bn.running_mean.clone(), bn.running_var.clone()))
    assert torch.equal(out[0][0], ref) and torch.equal(out[1][0], ref)
    assert torch.equal(out[0][5], out[1][5])                                  # min / max: bit for bit
    y64 = ref.double()
    mean64, var64 = y64.mean(dim=(0, 2, 3)), y64.var(dim=(0, 2, 3), unbiased=False)
    for o in out:
        assert float((o[1].double() - mean64).abs().max()) < 2e-6 * float(mean64.abs().max())
        assert float((o[2].double() * (var64 + 1e-5).sqrt() - 1).abs().max()) < 1e-5
    assert float((out[0][6] - out[1][6]).abs().max()) < 1e-6 * float(out[0][6].abs().max())
    assert float((out[0][7] / out[1][7] - 1).abs().max()) < 1e-5


def test_layers_without_such_a_kernel(bf16):
    """2-d planes, the small-layer kernels of conv_s1d.hip, other arithmetics: no layout, conv_forward leaves no records."""
    assert _layout(F._desc(128, 476, 476, 1, 3, 1, 3)) is None                # conv_s1d
    assert _layout(F._desc(16, 64, 64, 8, 100, 3, 3)) is None                 # 2-d
    assert _layout(F._desc(128, 64, 64, 1, 1723, 1, 3, 9)) is None            # nine-product arithmetic
    x = torch.randn(128, 476, 1, 3, device=DEV)
    w = torch.randn(476, 476, 1, 3, device=DEV) * 0.05
    F._PRESTATS.clear()
    F.conv_forward(x, w, None, stats_bn=(torch.nn.BatchNorm2d(476).to(DEV).train(), True))
    assert not F._PRESTATS
    rec = torch.empty(16, device=DEV)
    with pytest.raises(_lib.FscError):
        d = F._desc(128, 476, 476, 1, 3, 1, 3)
        F.call("fsc_conv_fwd_stats", C.byref(d), F.ptr(x), F.ptr(x), None, F.ptr(x), None, F.ptr(rec), F.stream_ptr())


# the entry convolutions of cfg 3's blocks 0 - 2 at batch 128, odd rows (a last position without a window), a partly filled batch
POOL_CASES = [(128, 129, 64, 3446), (128, 64, 80, 1723), (128, 80, 100, 861), (64, 64, 64, 1001), (32, 48, 64, 2050)]


@pytest.mark.parametrize("case", POOL_CASES)
def test_convolution_max_pool_and_statistics_in_one_launch(case, bf16):
    n, ci, co, length = case
    torch.manual_seed(sum(case))
    x = torch.randn(n, ci, 1, length, device=DEV)
    w = torch.randn(co, ci, 1, 3, device=DEV) / (ci * 3) ** 0.5
    b = torch.randn(co, device=DEV) * 2.0
    x[0, :, 0, 4:8] = 0.0                                      # (equal neighbours: the first position of a window wins)
    d = F._desc(n, ci, co, 1, length, 1, 3)
    assert _lib.load().fsc_conv_fwd_pool_stats_supported(C.byref(d)) == 1, F.plan_name(d, 0)
    full = F.conv_forward(x, w, b)
    p_ref, idx_ref = F.maxpool_forward(full, 1)
    bn1, bn2 = (torch.nn.BatchNorm2d(co).to(DEV).train() for _ in range(2))
    for bn in (bn1, bn2):
        with torch.no_grad():
            bn.running_mean.copy_(b + 0.5)
    F._PRESTATS.clear()
    got = F.conv_pool1d_forward(x, w, b, (bn1, True))
    assert got is not None and F._PRESTATS
    p, idx, c_shape = got
    assert c_shape == (n, co, 1, length) and tuple(p.shape) == (n, co, 1, length // 2)
    assert torch.equal(p, p_ref) and torch.equal(idx, idx_ref)
    st1 = F.bn_prepare(p, bn1, True)
    assert not F._PRESTATS
    st2 = F.bn_prepare(p_ref, bn2, True)
    assert torch.equal(st1.minmax, st2.minmax)
    assert float((st1.mean - st2.mean).abs().max()) < 2e-6 * float(st2.mean.abs().max())
    assert float((st1.invstd / st2.invstd - 1).abs().max()) < 1e-5
    assert float((bn1.running_var / bn2.running_var - 1).abs().max()) < 1e-5


def test_no_pool_fusion_where_the_layer_has_no_such_kernel(bf16):
    assert _lib.load().fsc_conv_fwd_pool_stats_supported(C.byref(F._desc(128, 381, 476, 1, 6, 1, 3))) == 0       # conv_s1d
    assert _lib.load().fsc_conv_fwd_pool_stats_supported(C.byref(F._desc(128, 64, 64, 1, 1723, 1, 1))) == 0      # 1 x 1
    x = torch.randn(128, 381, 1, 6, device=DEV)
    w = torch.randn(476, 381, 1, 3, device=DEV) * 0.03
    assert F.conv_pool1d_forward(x, w, None, (torch.nn.BatchNorm2d(476).to(DEV).train(), True)) is None
