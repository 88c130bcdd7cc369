"""GPU-box diagnostic: HIP engine vs a plain-PyTorch fp64 run of the same student step (not a test)."""
import copy, sys, os
from functools import reduce
import numpy as np, torch
from torch import nn
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import kdcc_amd
from kdcc_amd.models import DeepWV3Plus
from kdcc_amd.models.students import DepthwiseStudent
from kdcc_amd import losses
from _seeded import seeded_fill_, seeded_input

plan = ["mod4.block2.convs.conv2", "mod4.block3.convs.conv1", "mod7.block1.convs.conv2", "aspp.features.1.0"]
def get(m, name): return reduce(lambda a, e: a[int(e)] if e.isdigit() else getattr(a, e), name.split('.'), m)

teacher = DeepWV3Plus(19); seeded_fill_(teacher, "teacher."); teacher.eval()
model = DepthwiseStudent(teacher, None, dtype=torch.float32)
model.replace([{"name": n, "epoch": 1} for n in plan], kernel_size=9, padding=20, dilation=5)
model.register_hint_layers(plan); model.unfreeze(plan)
for n in plan: seeded_fill_(model.get_block(n, model.student), f"student.{n}.")
model.cuda()
x = seeded_input("step.g4.x", (2, 3, 64, 128)).cuda()
out_st, out_tc = model(x)
crit = losses.MSELoss(num_classes=1000)
hint = 0
for s, t in zip(model.student_hidden_outputs, model.teacher_hidden_outputs): hint = hint + crit(s, t)
hint.backward(); torch.cuda.synchronize()

# fp64 torch reference on the GPU
class DW(nn.Module):
    def __init__(s, b):
        super().__init__()
        s.separable_conv = nn.Conv2d(b.in_channels, b.in_channels, 9, padding=20, dilation=5, groups=b.in_channels, bias=False)
        s.pointwise_conv = nn.Conv2d(b.in_channels, b.out_channels, 1, bias=False)
        s.separable_conv.weight.data.copy_(b.separable_conv.weight.data); s.pointwise_conv.weight.data.copy_(b.pointwise_conv.weight.data)
    def forward(s, x): return s.pointwise_conv(s.separable_conv(x))
ref_t = copy.deepcopy(model.teacher).double()
ref_s = copy.deepcopy(model.student)
for n in plan:
    parts = n.split('.'); o = get(ref_s, '.'.join(parts[:-1])); b = DW(get(model.student, n))
    if parts[-1].isdigit(): o[int(parts[-1])] = b
    else: setattr(o, parts[-1], b)
ref_s = ref_s.double().cuda()
for p in ref_s.parameters(): p.requires_grad = False
for n in plan:
    for p in get(ref_s, n).parameters(): p.requires_grad = True
th, sh = [], []
for n in plan:
    def ht(m, i, o): th.append(o)
    def hs(m, i, o): sh.append(o)
    get(ref_t, n).register_forward_hook(ht)
    get(ref_s, n).register_forward_hook(hs)
xd = x.double()
with torch.no_grad(): tc = ref_t(xd)
st = ref_s(xd)
h2 = 0
for s, t in zip(sh, th): h2 = h2 + ((s - t) ** 2).mean() * 1000
h2.backward()
def err(a, b): return float((a.double() - b.double()).abs().max() / b.double().abs().max())
print("hint", hint.item(), h2.item())
print("logits st", err(out_st, st), "tc", err(out_tc, tc))
for i, n in enumerate(plan):
    print(n, "hint_s", err(model.student_hidden_outputs[i], sh[i]), "hint_t", err(model.teacher_hidden_outputs[i], th[i]))
for (n, p), (n2, p2) in zip([(n, p) for n, p in model.student.named_parameters() if p.requires_grad],
                            [(n, p) for n, p in ref_s.named_parameters() if p.requires_grad]):
    e = (p.grad.double() - p2.grad).abs()
    print(n, "grad err", float(e.max() / p2.grad.abs().max()), "argmax", np.unravel_index(int(e.argmax()), tuple(e.shape)),
          "mean rel", float(e.mean() / p2.grad.abs().mean()))
    if "mod7" in n and "pointwise" in n:
        ee = e.reshape(e.shape[0], -1)
        print("  per-co-block max:", [round(float(ee[i:i + 256].max() / p2.grad.abs().max()), 5) for i in range(0, ee.shape[0], 256)])
        print("  per-ci-block max:", [round(float(ee[:, i:i + 128].max() / p2.grad.abs().max()), 5) for i in range(0, ee.shape[1], 128)])
