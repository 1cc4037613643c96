"""Expected tilings of the L16 convolution kernels (fsc_conv_l16_fwd / _pool_fwd / _wgrad and their statistics variants) for the
shapes the tests and the benchmark run.  The planners are host code (they answer without a GPU), so these tables are checked on
the CPU (tests/test_l16_plans_cpu.py) and the GPU tests assert against them before they launch anything: a planner change that
moves a benchmark layer to another instantiation -- or drops its tiling -- fails a test instead of turning into a skip.

Reference layers: networks/classifiers.py:524-536 (block entry 3x3 + max-pool), :72-104 (1x1 / 3x3 / 1x1 residual unit).
"""


def cfg2_layers():
    """(c_in, c_out, h, w, k) of the 18 distinct convolution shapes (24 convolutions: conv1 and conv3 of a unit share theirs)
    of the cfg-2 model (BASELINE.json configs[1]: 6 blocks, base 100, growth 1.5, 128 x 431 log-mel image)."""
    out = []
    h, w, c_in = 128, 431, 2
    for depth in [int(1.5 ** k * 100) for k in range(6)]:
        out.append((c_in, depth, h, w, 3))
        h, w = h // 2, w // 2
        out += [(depth, depth, h, w, 1), (depth, depth, h, w, 3)]
        c_in = depth
    return out


F3, F1, WG = "conv_l16_fwd_kernel<3,3,%d,%d>", "conv_l16_fwd_kernel<1,1,%d,%d>", "conv_l16_wgrad_kernel<%d,%d,%d,%d>"

# layer -> (forward, input gradient, weight gradient, fused with the 2x2 max-pool?, statistics records layout of the plain kernel,
#           of the pooled kernel) at batch 128.  None = no L16 tiling: the layer runs on the fp32-input kernels of conv.hip.
CFG2_N128 = {
    (2, 100, 128, 431, 3): (None, None, None, False, None, None),                      # stem: direct fp32 kernels
    (100, 100, 64, 215, 1): (F1 % (7, 2), F1 % (7, 2), WG % (1, 1, 2, 4), None, (256, 1, 112, 0), None),
    (100, 100, 64, 215, 3): (F3 % (7, 2), F3 % (7, 2), WG % (3, 3, 4, 1), True, (256, 1, 112, 0), (256, 1, 112, 0)),
    (100, 150, 64, 215, 3): (F3 % (5, 2), F3 % (7, 2), WG % (3, 3, 4, 1), True, (256, 2, 80, 1), (256, 2, 80, 1)),
    (150, 150, 32, 107, 1): (F1 % (5, 2), F1 % (5, 2), WG % (1, 1, 3, 4), None, (256, 2, 80, 1), None),
    (150, 150, 32, 107, 3): (F3 % (5, 2), F3 % (5, 2), WG % (3, 3, 3, 1), True, (256, 2, 80, 1), (256, 2, 80, 1)),
    (150, 225, 32, 107, 3): (F3 % (8, 2), F3 % (5, 2), WG % (3, 3, 4, 1), True, (256, 2, 128, 1), (256, 2, 128, 1)),
    (225, 225, 16, 53, 1): (F1 % (8, 2), F1 % (8, 2), WG % (1, 1, 2, 4), None, (256, 2, 128, 1), None),
    (225, 225, 16, 53, 3): (F3 % (8, 2), F3 % (8, 2), WG % (3, 3, 4, 1), True, (256, 2, 128, 1), (256, 2, 128, 1)),
    (225, 337, 16, 53, 3): (F3 % (8, 2), F3 % (8, 2), WG % (3, 3, 4, 1), True, (255, 3, 128, 0), (255, 3, 128, 0)),
    (337, 337, 8, 26, 1): (F1 % (8, 2), F1 % (8, 2), WG % (1, 1, 3, 4), None, (192, 3, 128, 1), None),
    (337, 337, 8, 26, 3): (F3 % (8, 2), F3 % (8, 2), WG % (3, 3, 4, 1), False, (192, 3, 128, 1), None),
    (337, 506, 8, 26, 3): (F3 % (8, 2), F3 % (8, 2), WG % (3, 3, 4, 1), False, (256, 4, 128, 1), None),
    (506, 506, 4, 13, 1): (F1 % (8, 2), F1 % (8, 2), None, None, (128, 4, 128, 1), None),
    (506, 506, 4, 13, 3): (F3 % (8, 1), F3 % (8, 1), WG % (3, 3, 4, 1), False, (256, 4, 128, 1), None),
    (506, 759, 4, 13, 3): (F3 % (8, 1), F3 % (8, 1), WG % (3, 3, 4, 1), False, (192, 6, 128, 1), None),
    (759, 759, 2, 6, 1): (None, None, None, None, None, None),                          # 2 x 6 pixels: < 128 work items
    (759, 759, 2, 6, 3): (None, None, None, False, None, None),
}

# tests/test_l16_gpu.py CONV_CASES (n, c_in, c_out, h, w, k) -> (forward, input gradient) instantiation; the first ten are cfg-2
# layers at a reduced batch and must get the batch-128 instantiation
CONV_CASES = {
    (16, 100, 100, 64, 215, 3): (F3 % (7, 2), F3 % (7, 2)), (16, 100, 150, 64, 215, 3): (F3 % (5, 2), F3 % (7, 2)),
    (24, 150, 150, 32, 107, 3): (F3 % (5, 2), F3 % (5, 2)), (32, 150, 225, 32, 107, 3): (F3 % (8, 2), F3 % (5, 2)),
    (64, 225, 225, 16, 53, 3): (F3 % (8, 2), F3 % (8, 2)), (128, 337, 337, 8, 26, 3): (F3 % (8, 2), F3 % (8, 2)),
    (128, 506, 506, 4, 13, 3): (F3 % (8, 1), F3 % (8, 1)), (16, 100, 100, 64, 215, 1): (F1 % (7, 2), F1 % (7, 2)),
    (32, 150, 150, 32, 107, 1): (F1 % (5, 2), F1 % (5, 2)), (128, 225, 225, 16, 53, 1): (F1 % (8, 2), F1 % (8, 2)),
    (40, 33, 49, 17, 29, 3): (F3 % (4, 1), None), (40, 64, 48, 30, 31, 3): (F3 % (3, 2), F3 % (4, 2)),
    (36, 57, 130, 23, 40, 3): (F3 % (5, 2), F3 % (4, 2)), (64, 95, 64, 9, 77, 1): (F1 % (4, 2), None),
    (48, 127, 97, 12, 20, 3): (None, None),
}

# (n, c_in, c_out, h, w) of test_conv_l16_fused_with_maxpool -> (fused tiling?, statistics layout of the pooled kernel)
POOL_CASES = {
    (8, 100, 150, 64, 215): (True, (256, 2, 80, 1)), (16, 150, 225, 32, 107): (True, (224, 2, 128, 1)),
    (32, 225, 337, 16, 53): (True, (192, 3, 128, 1)), (64, 337, 506, 8, 26): (False, None), (24, 64, 96, 17, 43): (False, None),
    (12, 48, 80, 30, 64): (False, None), (128, 506, 759, 4, 13): (False, None), (16, 100, 150, 31, 107): (True, (224, 2, 80, 1)),
    (16, 100, 150, 33, 105): (True, (256, 2, 80, 1)), (40, 64, 96, 23, 45): (False, None),
}

# (n, c_in, c_out, h, w, k) of test_conv_epilogue_reduces_batchnorm_statistics -> statistics layout (None: no L16 tiling at all)
STAT_CASES = {
    (4, 100, 100, 64, 215, 3): (224, 1, 112, 0), (8, 150, 150, 32, 107, 3): (224, 2, 80, 1), (16, 225, 337, 16, 53, 3): (192, 3, 128, 1),
    (64, 337, 337, 8, 26, 3): (192, 3, 128, 1), (128, 506, 506, 4, 13, 3): (256, 4, 128, 1), (4, 100, 100, 64, 215, 1): (224, 1, 112, 0),
    (16, 150, 225, 32, 107, 1): (256, 2, 128, 1), (6, 64, 96, 17, 43, 3): None, (64, 64, 64, 32, 87, 1): None,
    (64, 64, 64, 32, 87, 3): (256, 1, 64, 0), (64, 96, 96, 16, 43, 3): (192, 1, 96, 0), (64, 64, 96, 32, 87, 3): (256, 1, 96, 0),
    (64, 96, 96, 16, 43, 1): (192, 1, 96, 0),
}


def plans(F, n, c_in, c_out, h, w, k):
    """What the library answers for one shape: (forward, input gradient, weight gradient, fused pool or None for 1x1, statistics
    layout, pooled statistics layout)."""
    import ctypes as C
    d = F._desc(n, c_in, c_out, h, w, k, k, 3)
    fwd = F.l16_plan_name(d, 0) if F.conv_l16_supported(d, 0) else None
    dg = F.l16_plan_name(d, 1) if F.conv_l16_supported(d, 1) else None
    wg = F.l16_wgrad_plan_name(d) if F.conv_l16_wgrad_supported(d) else None
    pool = bool(F._lib.load().fsc_conv_l16_pool_supported(C.byref(d))) if k == 3 else None
    return fwd, dg, wg, pool, F._stats_layout(d, False), (F._stats_layout(d, True) if pool else None)


def wgrad_batch(F, layer):
    """Smallest batch at which fsc_conv_l16_wgrad runs the instantiation batch 128 selects for this layer (the whole batch on
    planes up to 8 x 26, where the fp64 reference is cheap)."""
    c_in, c_out, h, w, k = layer
    if h * w <= 8 * 26:
        return 128
    want = F.l16_wgrad_plan_name(F._desc(128, c_in, c_out, h, w, k, k, 3))
    for n in (2, 4, 8, 16, 32, 64):
        d = F._desc(n, c_in, c_out, h, w, k, k, 3)
        if F.conv_l16_wgrad_supported(d) and F.l16_wgrad_plan_name(d) == want:
            return n
    return 128
