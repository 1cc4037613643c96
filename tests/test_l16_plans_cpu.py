"""The tilings the L16 planners choose (host code of libfsc_hip.so, no GPU needed) against the expected-support tables of
tests/l16_tables.py: every cfg-2 layer at batch 128 (the instantiations bench.py times), and every shape the GPU tests of
tests/test_l16_gpu.py run -- so "no tiling" can never hide behind a skipped GPU test."""
import pytest

from freesound_classification_amd import functional as F

import l16_tables as T


@pytest.mark.parametrize("layer", T.cfg2_layers(), ids=lambda l: "%dto%d_%dx%d_k%d" % l)
def test_cfg2_layer_plans_at_batch_128(layer):
    assert T.plans(F, 128, *layer) == T.CFG2_N128[layer]


def test_reduced_batch_cases_run_the_batch_128_instantiation():
    """The GPU parity tests run the large planes at a reduced batch: the kernel instantiation must be the batch-128 one."""
    for case, want in T.CONV_CASES.items():
        got = T.plans(F, *case)[:2]
        assert got == want, (case, got)
    for case in list(T.CONV_CASES)[:10]:
        n, c_in, c_out, h, w, k = case
        assert T.CONV_CASES[case] == T.CFG2_N128[(c_in, c_out, h, w, k)][:2], case
    for layer in T.cfg2_layers():
        want = T.CFG2_N128[layer][2]
        if want is not None:
            c_in, c_out, h, w, k = layer
            n = T.wgrad_batch(F, layer)
            assert F.l16_wgrad_plan_name(F._desc(n, c_in, c_out, h, w, k, k, 3)) == want, (layer, n)


def test_pooled_and_statistics_variants_of_the_gpu_test_shapes():
    for (n, c_in, c_out, h, w), (fused, lay) in T.POOL_CASES.items():
        p = T.plans(F, n, c_in, c_out, h, w, 3)
        assert p[0] is not None and p[3] == fused and p[5] == lay, ((n, c_in, c_out, h, w), p)
    for case, lay in T.STAT_CASES.items():
        p = T.plans(F, *case)
        assert p[4] == lay, (case, p)


def _describe(d, mode):
    import ctypes as C
    from freesound_classification_amd._lib import call
    buf = C.create_string_buffer(256)
    call("fsc_conv_plan_describe", C.byref(d), mode, buf, 256)
    return buf.value.decode()


def test_cfg3_late_blocks_weight_gradients_run_one_unit_per_workgroup():
    """The 1-d model's last blocks (128 x 3 ... 13 positions: 7 - 32 boxes of 64 pixels) split their weight gradients down to one
    unit per workgroup on the native kernels (fp32 arithmetic); layers of 64 units and more keep at least four units per split
    (conv.hip plan_wgrad).  In bf16 (arith 1, cfg 3) these layers take conv_s1d.hip since round 6: 32-position K steps, split so
    that a wave keeps at least two steps."""
    want = {(476, 476, 3, 1): (7, 7), (476, 476, 3, 3): (7, 7), (381, 381, 6, 3): (13, 13), (305, 305, 13, 1): (32, 32),
            (244, 244, 26, 1): (64, 16)}
    for (ci, co, length, k), (units, split) in want.items():
        txt = _describe(F._desc(128, ci, co, 1, length, 1, k, 0), 2)
        assert txt.startswith("conv_wgrad_kernel") and "units=%d split=%d " % (units, split) in txt, txt
    # (rows of >= 13 frames: K steps of four 8-position segments of single rows -- 128 x ceil(L / 8) / 4 units; shorter rows: 32
    # flattened positions per unit)
    s1d = {(476, 476, 3, 1): (12, 1), (476, 476, 3, 3): (12, 1), (381, 381, 6, 3): (24, 3), (305, 305, 13, 1): (64, 6),
           (244, 244, 26, 1): (128, 8), (195, 195, 53, 3): (224, 11), (156, 156, 107, 3): (448, 21)}
    for (ci, co, length, k), (units, split) in s1d.items():
        d = F._desc(128, ci, co, 1, length, 1, k, 1)
        txt = _describe(d, 2)
        assert txt.startswith("conv_s1d_wgrad_kernel<1,%d>" % k) and "units=%d split=%d " % (units, split) in txt, txt
        assert _describe(d, 0).startswith("conv_s1d_fwd_kernel<1,%d>" % k) and _describe(d, 1).startswith("conv_s1d_fwd_kernel")
    # the ranges: forward and k = 3 input gradient up to 32 768 positions, the k = 1 input gradient and the weight gradient's segment
    # form up to 65 536 (the k = 1 FORWARD at 55 k positions is the ring kernel's again: it has the statistics epilogue); rows above
    # keep the round-1 kernels in every direction
    d = F._desc(128, 125, 156, 1, 215, 1, 3, 1)
    assert _describe(d, 0).startswith("conv_s1d_fwd_kernel") and _describe(d, 2).startswith("conv_s1d_wgrad_kernel")
    d = F._desc(128, 100, 100, 1, 430, 1, 3, 1)
    assert _describe(d, 0).startswith("conv_fwd_x3_kernel") and _describe(d, 2).startswith("conv_s1d_wgrad_kernel")
    d = F._desc(128, 100, 100, 1, 430, 1, 1, 1)
    assert _describe(d, 0).startswith("conv_fwd_x3_kernel") and _describe(d, 1).startswith("conv_s1d_fwd_kernel")
    assert _describe(d, 2).startswith("conv_s1d_wgrad_kernel")
    d = F._desc(128, 80, 80, 1, 861, 1, 1, 1)
    assert _describe(d, 0).startswith("conv_fwd_x3_kernel") and _describe(d, 2).startswith("conv_wgrad_kernel")


def test_multi_pack_covers_the_bf16_limb_tilings_only():
    lib = F._lib.load()
    import ctypes as C
    for (ci, co, length, k) in [(129, 64, 3446, 3), (100, 125, 430, 3), (476, 476, 3, 1)]:
        for dgrad in (0, 1):
            assert lib.fsc_conv_pack_weights_multi_supported(C.byref(F._desc(128, ci, co, 1, length, 1, k, 1)), dgrad)
    assert not lib.fsc_conv_pack_weights_multi_supported(C.byref(F._desc(128, 759, 759, 2, 6, 3, 3, 3)), 0)     # scaled fp16 limbs
    assert not lib.fsc_conv_pack_weights_multi_supported(C.byref(F._desc(128, 759, 759, 2, 6, 3, 3, 0)), 0)     # native fp32 tiling
