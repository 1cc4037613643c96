"""development: where the deviation of the first block's input-BatchNorm gradients (conv_modules.0.0.weight / .bias) comes
from.  Block 0 of the cfg-2 model on a golden-like batch: dgamma / dbeta through (a) the weight-gradient identity of DESIGN 4.5
(STEM_BN_IDENTITY, shipped), (b) the explicit route (stem input gradient + BatchNorm backward), against the CPU oracle in fp32 and
fp64, together with the conditioning of the identity's final contraction sum_{co,tap} w dW'."""
import os, sys, copy
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from freesound_classification_amd import functional as F
from freesound_classification_amd.networks.classifiers import TwoDimensionalCNNClassificationModel
from oracle import ref_torch as oref
import torch.nn.functional as TF
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from test_cfg2_gpu import cfg2_experiment

DEV = torch.device("cuda:0")
for arith in ("bf16x9", "f16x3"):
    F.set_conv_arith(arith)
    torch.manual_seed(int(sys.argv[sys.argv.index('--seed') + 1]) if '--seed' in sys.argv else 3)
    m = TwoDimensionalCNNClassificationModel(cfg2_experiment(), device="cuda:0")
    state = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    ref = oref.TagCNN2d("mel_2048_1024_128", 6, 100, 1.5, 1, 80)
    ref.load_state_dict(state)
    ref.train(); m.train()
    gen = torch.Generator().manual_seed(17 if '--seed' in sys.argv else 5)
    signal = 0.1 * torch.randn(4, 441000, 1, generator=gen)
    if "--golden" in sys.argv:             # the batch of fixture g12 (tests/test_cfg2_gpu.py): rows with zero-padded tails, one quiet row
        import numpy as np
        from test_oracle_cpu import cfg2_golden_inputs
        with np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "g12_cfg2_step.npz")) as z:
            signal, _ = cfg2_golden_inputs({k: z[k] for k in z.files})
    with torch.no_grad():
        x = ref.front_end(signal)
    blk, mods = ref.conv_modules[0], m.conv_modules[0]
    g_out = 1e-3 * torch.randn(signal.shape[0], 100, 64, 215, generator=gen)

    def oracle(dtype):
        b = copy.deepcopy(blk).to(dtype)
        for q in b.parameters():
            q.grad = None
        out = b(x.to(dtype))
        out.backward(g_out.to(dtype))
        return {k: q.grad.double() for k, q in b.named_parameters()}

    g64, g32 = oracle(torch.float64), oracle(torch.float32)
    res = {}
    for ident in (True, False):
        F.STEM_BN_IDENTITY = ident
        for p in mods.parameters():
            p.grad = None
        xp = x.to(DEV).requires_grad_(not ident)
        out_p, _ = F.conv_block(xp, mods, True, False, 2)
        out_p.backward(g_out.to(DEV))
        res[ident] = {k: p.grad.detach().cpu().double() for k, p in mods.named_parameters()}
    F.STEM_BN_IDENTITY = True
    for k in ("0.weight", "0.bias", "1.weight"):
        scale = max(1.0, float(g64[k].abs().max()))
        print(arith, k, "scale %.3g | vs fp64: oracle fp32 %.2e, identity route %.2e, explicit route %.2e | value %s" % (
            scale, float((g32[k] - g64[k]).abs().max()) / scale, float((res[True][k] - g64[k]).abs().max()) / scale,
            float((res[False][k] - g64[k]).abs().max()) / scale, g64[k].flatten()[:2].tolist()))
    # conditioning of the final contraction: sum |w dW| against |sum w dW| per input channel
    xs = x.double()
    print("   input channel 0: mean %.4f std %.4f min %.4f max %.4f; share of positions at log(1e-4): %.3f" % (
        float(xs[:, 0].mean()), float(xs[:, 0].std()), float(xs[:, 0].min()), float(xs[:, 0].max()),
        float((xs[:, 0] < -9.2).double().mean())))
    w = state["conv_modules.0.1.weight"].double(); dw = g64["1.weight"]
    for ci in range(2):
        t = (w[:, ci] * dw[:, ci])
        print("   channel %d: sum |w dW| = %.3e, |sum w dW| = %.3e" % (ci, float(t.abs().sum()), abs(float(t.sum()))))
F.set_conv_arith(None)
