"""Development: eager vs replayed training steps of a small 1-d model (what tests/test_r4_gpu.py asserts), with the differences printed."""
import copy, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from freesound_classification_amd import functional as F
from freesound_classification_amd.ops.training import CapturedTrainingStep, make_step
from test_r4_gpu import _small_1d, DEV

F.set_conv_arith(sys.argv[1] if len(sys.argv) > 1 else "f16x3")
torch.manual_seed(5)
_side = torch.cuda.Stream(device=DEV); _side.wait_stream(torch.cuda.current_stream(DEV)); torch.cuda.set_stream(_side)   # one non-default stream for everything
model = _small_1d(weight_decay=float(sys.argv[2]) if len(sys.argv) > 2 else 0.0)
model.train(); model.global_step = 0
model.make_optimizer(max_steps=12)
gen = torch.Generator(device=DEV).manual_seed(3)
batches = [(0.1 * torch.randn(16, 44100, 1, device=DEV, generator=gen), (torch.rand(16, 80, device=DEV, generator=gen) < 0.05).float()) for _ in range(4)]
model.global_step += 1; make_step(model.scheduler, step=model.global_step)
model.training_step(*batches[0])
state = copy.deepcopy(model.state_dict()); ostate = copy.deepcopy(model.optimizer.state_dict()); step0 = model.global_step; epoch0 = model.scheduler.epoch
def restore():
    model.load_state_dict(state); model.optimizer.load_state_dict(copy.deepcopy(ostate))
def run(fn):
    model.global_step = step0; model.scheduler.epoch = epoch0; out = []
    for x, y in batches:
        model.global_step += 1; make_step(model.scheduler, step=model.global_step)
        out.append(fn(x, y)[0].detach().clone())
    torch.cuda.synchronize()
    return out, copy.deepcopy(model.state_dict())
restore(); e1, s1 = run(model.training_step)
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
ee = []
for _ in range(reps):
    restore(); e2, s2 = run(model.training_step)
    ee.append(max(float((a - b).abs().max()) for a, b in zip(e1, e2)))
print("eager vs eager, worst logit difference over 4 steps, %d repetitions:" % reps, " ".join("%.1e" % v for v in sorted(ee)))
restore(); cap = CapturedTrainingStep(model, *batches[0])
saved = {p_: {k: (v.clone() if torch.is_tensor(v) else v) for k, v in st.items()} for p_, st in model.optimizer.state.items()}
count0 = cap.step_count
er = []
for _ in range(reps):
    model.load_state_dict(state)                            # in place
    for p_, st in model.optimizer.state.items():            # in place: the graph holds these addresses
        for k, v in saved[p_].items():
            if torch.is_tensor(v):
                st[k].copy_(v)
            else:
                st[k] = v
    cap.step_count = count0
    r1, t1 = run(cap)
    er.append(max(float((a - b).abs().max()) for a, b in zip(e1, r1)))
print("eager vs replay:", " ".join("%.1e" % v for v in sorted(er)))
