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
