#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE itself (CPU, fp32).

Runs only in the build container (it imports /root/reference, which does not
exist on the GPU box and must never travel).  Only data -- seeded inputs and
the reference's outputs -- is written; no reference source is copied.

    cd /root/reference && python3 /root/repo/tools/make_golden.py [--only NAME]

Import recipe: SURVEY.md Appendix C (stub the seven missing third-party
imports, neutralise the hard-coded .cuda() calls).
"""
import argparse
import math
import os
import sys
import types

sys.dont_write_bytecode = True
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("KD_REFERENCE", "/root/reference")
os.chdir(REF)
sys.path.insert(0, REF)
sys.path.insert(0, os.path.join(REPO, "tests"))

import numpy as np
import torch
from torch import nn


class _Any:
    def __init__(s, *a, **k): pass
    def __call__(s, *a, **k): return s
    def __getattr__(s, n): return _Any()
    def __setitem__(s, k, v): pass
    def __getitem__(s, k): return _Any()
    def __str__(s): return "<stub>"


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    m.__path__ = []
    sys.modules[name] = m
    return m


_stub("beautifultable", BeautifulTable=_Any)
tv = _stub("torchvision")
tv.transforms = _stub("torchvision.transforms", ToPILImage=_Any, Compose=_Any, ToTensor=_Any, Normalize=_Any, Lambda=_Any)
tv.utils = _stub("torchvision.utils", make_grid=_Any())
tv.datasets = _stub("torchvision.datasets", CIFAR10=_Any, CIFAR100=_Any)
_stub("torchvision.datasets.utils", extract_archive=_Any(), verify_str_arg=_Any(), iterable_to_str=_Any())
_stub("cv2"); _stub("tensorboardX", SummaryWriter=_Any); _stub("torchsummary", summary=_Any())
_stub("skimage"); _stub("skimage.filters", gaussian=_Any()); _stub("skimage.restoration", denoise_bilateral=_Any())
_stub("torch._six", inf=math.inf, string_classes=(str,))
nn.Module.cuda = lambda self, *a, **k: self
torch.cuda.empty_cache = lambda: None

import warnings
warnings.filterwarnings("ignore")

import losses as ref_losses                                            # noqa: E402
import models as ref_models                                            # noqa: E402
from models.students import DepthwiseStudent                           # noqa: E402
from models.students.transform_blocks import DepthwiseSeparableBlock   # noqa: E402
from models.encoders.wider_resnet import IdentityResidualBlock, bnrelu  # noqa: E402
from models.deeplabv3.deeplabv3 import _AtrousSpatialPyramidPoolingModule, DeepWV3Plus  # noqa: E402
from utils.optim.radam import RAdam                                    # noqa: E402

from _seeded import seeded_fill_, seeded_input, summarize              # noqa: E402

OUT = os.path.join(REPO, "tests", "golden")
torch.set_num_threads(8)


def save(name, **arrs):
    flat = {}
    for k, v in arrs.items():
        if isinstance(v, dict):
            for kk, vv in v.items():
                flat[f"{k}.{kk}"] = np.asarray(vv)
        elif torch.is_tensor(v):
            flat[k] = v.detach().cpu().numpy()
        else:
            flat[k] = np.asarray(v)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **flat)
    print(f"wrote {path}  ({os.path.getsize(path)/1024:.1f} KiB)")


# ----------------------------------------------------------------------------
def g_losses():
    """losses/KLDiv.py, MSELoss.py, WeightedHintMSELoss.py, CrossEntropy.py: (inputs, targets) -> (loss, grad)."""
    out = {}

    def run(tag, crit, s, t, *extra):
        s = s.clone().requires_grad_(True)
        loss = crit(s, t, *extra)
        loss.backward()
        out[f"{tag}.s"] = s.detach().numpy()
        out[f"{tag}.t"] = t.numpy()
        out[f"{tag}.loss"] = np.float64(loss.item())
        out[f"{tag}.grad"] = s.grad.numpy()

    s4, t4 = seeded_input("loss.s4", (2, 19, 8, 16)), seeded_input("loss.t4", (2, 19, 8, 16))
    run("kld_T1", ref_losses.KLDivergenceLoss(1), s4, t4)
    run("kld_T5", ref_losses.KLDivergenceLoss(5), s4 * 3, t4 * 3)
    s2, t2 = seeded_input("loss.s2", (32, 10), 2.0), seeded_input("loss.t2", (32, 10), 2.0)
    run("kld2d_T5", ref_losses.KLDivergenceLoss(5), s2, t2)
    h, g = seeded_input("loss.h", (2, 24, 8, 16)), seeded_input("loss.g", (2, 24, 8, 16))
    run("mse_1000", ref_losses.MSELoss(num_classes=1000), h, g)
    run("mse_1", ref_losses.MSELoss(num_classes=1), h, g)
    wc = torch.rand(24, generator=torch.Generator().manual_seed(7))
    wnc = torch.rand(2, 24, generator=torch.Generator().manual_seed(8))
    run("whmse_c", ref_losses.WeightedHintMSELoss(), h, g, wc)
    out["whmse_c.w"] = wc.numpy()
    run("whmse_nc", ref_losses.WeightedHintMSELoss(), h, g, wnc)
    out["whmse_nc.w"] = wnc.numpy()
    # CrossEntropyLoss2d (logged metric), with an ignore band
    tgt = torch.randint(0, 19, (2, 8, 16), generator=torch.Generator().manual_seed(9))
    tgt[:, :2] = 255
    ce = ref_losses.CrossEntropyLoss2d(ignore_index=255)(s4, tgt)
    out["ce.x"] = s4.numpy(); out["ce.target"] = tgt.numpy(); out["ce.loss"] = np.float64(ce.item())
    # the SURVEY 8c anchors (torch.manual_seed(0); randn twice)
    torch.manual_seed(0)
    x, t = torch.randn(2, 19, 8, 16), torch.randn(2, 19, 8, 16)
    out["anchor.kld_T1"] = np.float64(ref_losses.KLDivergenceLoss(1)(x, t).item())
    out["anchor.kld_T5"] = np.float64(ref_losses.KLDivergenceLoss(5)(x, t).item())
    out["anchor.mse_1000"] = np.float64(ref_losses.MSELoss(num_classes=1000)(x, t).item())
    out["anchor.x"] = x.numpy(); out["anchor.t"] = t.numpy()
    save("losses", **out)


def g_dwsep():
    """models/students/transform_blocks/depthwise_separable_conv.py: fwd + all grads."""
    out = {}
    for tag, (C, Co, k, p, d, H, W) in {
        "k9d5": (16, 24, 9, 20, 5, 24, 32),
        "k3d1": (16, 16, 3, 1, 1, 8, 8),
    }.items():
        blk = DepthwiseSeparableBlock(C, Co, k, p, d, groups=C, bias=False)
        seeded_fill_(blk, f"dwsep.{tag}.")
        x = seeded_input(f"dwsep.{tag}.x", (2, C, H, W)).requires_grad_(True)
        y = blk(x)
        gy = seeded_input(f"dwsep.{tag}.gy", tuple(y.shape))
        y.backward(gy)
        out[f"{tag}.cfg"] = np.array([C, Co, k, p, d, H, W])
        out[f"{tag}.x"] = x.detach().numpy()
        out[f"{tag}.w_dw"] = blk.separable_conv.weight.detach().numpy()
        out[f"{tag}.w_pw"] = blk.pointwise_conv.weight.detach().numpy()
        out[f"{tag}.y"] = y.detach().numpy()
        out[f"{tag}.gy"] = gy.numpy()
        out[f"{tag}.gx"] = x.grad.numpy()
        out[f"{tag}.gw_dw"] = blk.separable_conv.weight.grad.numpy()
        out[f"{tag}.gw_pw"] = blk.pointwise_conv.weight.grad.numpy()
    save("dwsep", **out)


def g_variants():
    """Constructor variants the shipped KD configs do not use but the reference's classes accept: MSELoss(reduction='sum')
    (losses/MSELoss.py:9-16), CrossEntropyLoss2d(weight, size_average) (losses/CrossEntropy.py:5-14), and
    DepthwiseSeparableBlock(bias=True) (depthwise_separable_conv.py:7-9): values and every gradient."""
    out = {}
    h, g = seeded_input("var.h", (2, 24, 9, 13)), seeded_input("var.g", (2, 24, 9, 13))
    s = h.clone().requires_grad_(True)
    loss = ref_losses.MSELoss(reduction='sum', num_classes=19)(s, g)
    loss.backward()
    out.update({"mse_sum.s": h.numpy(), "mse_sum.t": g.numpy(), "mse_sum.loss": np.float64(loss.item()), "mse_sum.grad": s.grad.numpy()})
    x = seeded_input("var.ce.x", (2, 19, 11, 17), 2.0)
    tgt = torch.randint(0, 19, (2, 11, 17), generator=torch.Generator().manual_seed(11))
    tgt[0, :2] = 255
    w = torch.rand(19, generator=torch.Generator().manual_seed(12)) + 0.25
    out.update({"ce.x": x.numpy(), "ce.target": tgt.numpy(), "ce.w": w.numpy()})
    for tag, kw in (("ce_w_mean", dict(weight=w, size_average=True)), ("ce_w_sum", dict(weight=w, size_average=False)),
                    ("ce_sum", dict(size_average=False))):
        xs = x.clone().requires_grad_(True)
        loss = ref_losses.CrossEntropyLoss2d(ignore_index=255, **kw)(xs, tgt)
        loss.backward()
        out[f"{tag}.loss"] = np.float64(loss.item())
        out[f"{tag}.grad"] = xs.grad.numpy()
    C, Co, k, p, d, H, W = 64, 64, 3, 2, 2, 12, 16
    blk = DepthwiseSeparableBlock(C, Co, k, p, d, groups=C, bias=True)
    seeded_fill_(blk, "var.dwsep_bias.")
    xb = seeded_input("var.dwsep_bias.x", (2, C, H, W)).requires_grad_(True)
    y = blk(xb)
    gy = seeded_input("var.dwsep_bias.gy", tuple(y.shape))
    y.backward(gy)
    out["dwsep_bias.cfg"] = np.array([C, Co, k, p, d, H, W])
    for n, t in (("x", xb.detach()), ("w_dw", blk.separable_conv.weight.detach()), ("b_dw", blk.separable_conv.bias.detach()),
                 ("w_pw", blk.pointwise_conv.weight.detach()), ("b_pw", blk.pointwise_conv.bias.detach()), ("y", y.detach()), ("gy", gy),
                 ("gx", xb.grad), ("gw_dw", blk.separable_conv.weight.grad), ("gb_dw", blk.separable_conv.bias.grad),
                 ("gw_pw", blk.pointwise_conv.weight.grad), ("gb_pw", blk.pointwise_conv.bias.grad)):
        out[f"dwsep_bias.{n}"] = t.numpy()
    save("variants", **out)


def g_resblock():
    """models/encoders/wider_resnet.py:119-182 IdentityResidualBlock in eval mode: fwd + input grad."""
    out = {}
    cases = {
        "id2_d1": dict(cin=16, ch=(16, 16), stride=1, dil=1),           # identity shortcut, 2-conv
        "proj2_s2": dict(cin=8, ch=(16, 16), stride=2, dil=1),          # proj + stride 2 (mod4.block1 shape class)
        "proj2_d2": dict(cin=16, ch=(8, 32), stride=1, dil=2),          # mod5 shape class
        "bott_d4": dict(cin=16, ch=(8, 16, 32), stride=1, dil=4),       # mod6/mod7 bottleneck
    }
    for tag, c in cases.items():
        blk = IdentityResidualBlock(c["cin"], c["ch"], stride=c["stride"], dilation=c["dil"], norm_act=bnrelu)
        seeded_fill_(blk, f"resblock.{tag}.")
        blk.eval()
        x = seeded_input(f"resblock.{tag}.x", (2, c["cin"], 12, 16)).requires_grad_(True)
        y = blk(x)
        gy = seeded_input(f"resblock.{tag}.gy", tuple(y.shape))
        y.backward(gy)
        out[f"{tag}.cfg"] = np.array([c["cin"], c["stride"], c["dil"], len(c["ch"])] + list(c["ch"]))
        out[f"{tag}.x"] = x.detach().numpy(); out[f"{tag}.y"] = y.detach().numpy()
        out[f"{tag}.gy"] = gy.numpy(); out[f"{tag}.gx"] = x.grad.numpy()
    save("resblock", **out)


def g_aspp():
    """models/deeplabv3/deeplabv3.py:21-75 ASPP (output_stride 8 -> rates 12/24/36), eval mode."""
    aspp = _AtrousSpatialPyramidPoolingModule(32, 16, output_stride=8)
    seeded_fill_(aspp, "aspp.")
    aspp.eval()
    x = seeded_input("aspp.x", (2, 32, 16, 24))
    with torch.no_grad():
        y = aspp(x)
    save("aspp", x=x.numpy(), y=y.numpy())


def g_ops():
    """Functional ops the trunk uses: max-pool 3x3/s2/p1, bilinear align_corners upsample, global avg-pool."""
    x = seeded_input("ops.x", (2, 6, 11, 14))
    mp = nn.MaxPool2d(3, stride=2, padding=1)(x)
    up = nn.functional.interpolate(x, size=(29, 40), mode="bilinear", align_corners=True)
    up2 = nn.functional.interpolate(x, size=(22, 28), mode="bilinear", align_corners=True)
    gp = nn.AdaptiveAvgPool2d(1)(x)
    # conv variants incl. stride 2 and dilation, plus grads
    w = seeded_input("ops.w", (8, 6, 3, 3)) * 0.2
    outs = {}
    for tag, (s, p, d) in {"s1d1": (1, 1, 1), "s2d1": (2, 1, 1), "s1d4": (1, 4, 4)}.items():
        xi = x.clone().requires_grad_(True); wi = w.clone().requires_grad_(True)
        y = nn.functional.conv2d(xi, wi, None, s, p, d)
        gy = seeded_input(f"ops.gy.{tag}", tuple(y.shape))
        y.backward(gy)
        outs[f"conv_{tag}.y"] = y.detach().numpy(); outs[f"conv_{tag}.gy"] = gy.numpy()
        outs[f"conv_{tag}.gx"] = xi.grad.numpy(); outs[f"conv_{tag}.gw"] = wi.grad.numpy()
    # train-mode BN (+ReLU) fwd/bwd: the ClassificationTrainer path (bn.training == True)
    bn = nn.BatchNorm2d(6); seeded_fill_(bn, "ops.bn."); bn.train()
    xb = x.clone().requires_grad_(True)
    yb = torch.relu(bn(xb))
    gyb = seeded_input("ops.gyb", tuple(yb.shape))
    yb.backward(gyb)
    save("ops", x=x.numpy(), w=w.numpy(), maxpool=mp.numpy(), up=up.numpy(), up2=up2.numpy(), gap=gp.numpy(),
         bn_gamma=bn.weight.detach().numpy(), bn_beta=bn.bias.detach().numpy(), bn_y=yb.detach().numpy(),
         bn_gy=gyb.numpy(), bn_gx=xb.grad.numpy(), bn_ggamma=bn.weight.grad.numpy(), bn_gbeta=bn.bias.grad.numpy(),
         **outs)


def g_radam():
    """utils/optim/radam.py:30-98: 8 steps (crosses the N_sma>=5 switch at step 6) on one tensor."""
    p = seeded_input("radam.p", (64,)).requires_grad_(True)
    opt = RAdam([p], lr=0.005)
    ps, gs = [p.detach().clone().numpy()], []
    for i in range(8):
        g = seeded_input(f"radam.g{i}", (64,))
        p.grad = g.clone()
        opt.step()
        gs.append(g.numpy()); ps.append(p.detach().clone().numpy())
    save("radam", p=np.stack(ps), g=np.stack(gs), lr=0.005)


PLANS = {
    # 4 replaced blocks covering every hint class: aliased conv2 (F7), raw conv1, bottleneck conv2, ASPP branch
    "g4": ["mod4.block2.convs.conv2", "mod4.block3.convs.conv1", "mod7.block1.convs.conv2", "aspp.features.1.0"],
}


def build_student(plan, kernel=(9, 20, 5)):
    teacher = DeepWV3Plus(num_classes=19)
    seeded_fill_(teacher, "teacher.")
    teacher.eval()
    model = DepthwiseStudent(teacher, config=None)
    k, p, d = kernel
    model.replace([{"name": n, "epoch": 1} for n in plan], kernel_size=k, padding=p, dilation=d)
    model.register_hint_layers(plan)
    model.unfreeze(plan)
    for n in plan:  # new blocks: seeded by checkpoint key name
        seeded_fill_(model.get_block(n, model.student), f"student.{n}.")
    return model


def g_student_step(plan_name="g4", hw=(64, 128), batch=2):
    """DepthwiseStudent.forward + the four criteria + loss.backward() of
    trainer/layerwise_trainer.py:223-235 on seeded weights / inputs."""
    plan = PLANS[plan_name]
    model = build_student(plan)
    x = seeded_input(f"step.{plan_name}.x", (batch, 3) + hw)
    tgt = torch.randint(0, 19, (batch,) + hw, generator=torch.Generator().manual_seed(11))
    tgt[:, :4] = 255
    out_st, out_tc = model(x)
    crit = [ref_losses.CrossEntropyLoss2d(ignore_index=255), ref_losses.KLDivergenceLoss(1),
            ref_losses.MSELoss(num_classes=1000), ref_losses.MSELoss(num_classes=1)]
    sup = crit[0](out_st, tgt); kd = crit[1](out_st, out_tc); tl = crit[0](out_tc, tgt); kd_mse = crit[3](out_st, out_tc)
    hint = 0
    per_hint = []
    for s, t in zip(model.student_hidden_outputs, model.teacher_hidden_outputs):
        l = crit[2](s, t); per_hint.append(l.item()); hint = hint + l
    hint.backward()
    out = dict(plan=np.array(plan), x_key=f"step.{plan_name}.x", target=tgt.numpy().astype(np.uint8),
               hint_loss=np.float64(hint.item()), per_hint=np.array(per_hint), kd_loss=np.float64(kd.item()),
               kd_mse=np.float64(kd_mse.item()), supervised_loss=np.float64(sup.item()), teacher_loss=np.float64(tl.item()),
               student_logits=summarize(out_st), teacher_logits=summarize(out_tc))
    for i, (s, t) in enumerate(zip(model.student_hidden_outputs, model.teacher_hidden_outputs)):
        out[f"hint_s{i}"] = summarize(s); out[f"hint_t{i}"] = summarize(t)
    names = []
    for n, p in model.student.named_parameters():
        if p.requires_grad:
            names.append(n); out[f"grad:{n}"] = summarize(p.grad)
    out["trainable"] = np.array(names)
    save(f"student_step_{plan_name}", **out)


FULL_HINTS = ["mod4.block2.convs", "mod4.block3.convs.conv1", "mod7.block1", "aspp"]


def g_student_step_full(plan_name="g4", hw=(64, 128), batch=2):
    """SURVEY 8(d) mode B on the reference itself: the g4 student with EVERY student parameter trainable (what
    prepare_train_epoch's identical-architecture branch and `pruning.unfreeze` on dense blocks do,
    trainer/layerwise_trainer.py:88-100, models/students/depthwise_student.py:80-84; the student stays in eval mode),
    hints on a `convs` Sequential, a raw conv, a whole block and the ASPP module (the names of
    cfg/cityscapes/51M_deeplab_incremental.json), loss = KLDivergenceLoss(1)(student, teacher logits) + sum of hint MSEs,
    loss.backward().  Stores the losses and a summary of every parameter's gradient."""
    plan = PLANS[plan_name]
    model = build_student(plan)
    model.register_hint_layers(FULL_HINTS)
    for p in model.student.parameters():
        p.requires_grad = True
    x = seeded_input(f"step.{plan_name}.x", (batch, 3) + hw)
    out_st, out_tc = model(x)
    kd = ref_losses.KLDivergenceLoss(1)(out_st, out_tc)
    crit = ref_losses.MSELoss(num_classes=1000)
    hint, per = 0, []
    for s, t in zip(model.student_hidden_outputs, model.teacher_hidden_outputs):
        l = crit(s, t); per.append(l.item()); hint = hint + l
    (kd + hint).backward()
    out = dict(plan=np.array(plan), hints=np.array(FULL_HINTS), x_key=f"step.{plan_name}.x", kd_loss=np.float64(kd.item()),
               hint_loss=np.float64(hint.item()), per_hint=np.array(per), student_logits=summarize(out_st, 1024))
    for i, s in enumerate(model.student_hidden_outputs):
        out[f"hint_s{i}"] = summarize(s, 512)
    names = []
    for n, p in model.student.named_parameters():
        assert p.grad is not None, n
        names.append(n); out[f"grad:{n}"] = summarize(p.grad, 256)
    out["trainable"] = np.array(names)
    save(f"student_step_full_{plan_name}", **out)


TAYLOR_GATES = [("mod4.block2.convs.conv2", 512), ("mod7.block1.convs.conv2", 2048), ("aspp.features.1.0", 256)]


def g_taylor(hw=(64, 128), batch=2):
    """models/students/taylor_prune_student.py + trainer/taylor_prune_trainer.py:196-211 on the reference: gates behind the last
    conv of a residual block, a bottleneck's middle conv and an ASPP branch conv; loss = CrossEntropyLoss2d(student logits,
    target); importance = (gate * d loss / d gate)^2 from get_gate_importance(); ImportanceFilterTracker.average()."""
    from models.students.taylor_prune_student import TaylorPruneStudent
    from utils.util import ImportanceFilterTracker
    teacher = DeepWV3Plus(num_classes=19)
    seeded_fill_(teacher, "teacher.")
    teacher.eval()
    model = TaylorPruneStudent(teacher, config=None)
    model.replace([{"name": n, "epoch": 1, "num_features": c} for n, c in TAYLOR_GATES])
    x = seeded_input("taylor.x", (batch, 3) + hw)
    tgt = torch.randint(0, 19, (batch,) + hw, generator=torch.Generator().manual_seed(21))
    tgt[:, :4] = 255
    out_st, _ = model(x)
    loss = ref_losses.CrossEntropyLoss2d(ignore_index=255)(out_st, tgt)
    loss.backward()
    imp = model.get_gate_importance()
    tr = ImportanceFilterTracker(writer=None)
    tr.update_importance_list(model.added_gates)
    tr.update(imp)
    avg = tr.average()
    out = dict(names=np.array([n for n, _ in TAYLOR_GATES]), target=tgt.numpy().astype(np.uint8), loss=np.float64(loss.item()))
    for n, _ in TAYLOR_GATES:
        out[f"imp:{n}"] = np.asarray(imp[n], np.float64)
        out[f"gate_grad:{n}"] = model.added_gates[n].weight.grad.numpy()
        out[f"avg:{n}"] = np.asarray(avg[n], np.float64)
    save("taylor", **out)


TAYLOR_STEP_GATES = [("mod4.block2.convs.conv2", 512),      # behind the last conv of a block (the shortcut joins after the gate)
                     ("mod4.block3.convs.bn2.1", 512),       # behind the ReLU of a bnrelu inside a block
                     ("mod7.block1.convs.bn3.1", 2048),      # bottleneck block, second bnrelu
                     ("aspp.features.1.2", 256),             # behind an ASPP branch's ReLU
                     ("aspp.features.2.0", 256)]             # behind an ASPP branch's conv


def g_taylor_steps(hw=(64, 128), batch=2, steps=3, lr=0.05):
    """trainer/taylor_prune_trainer.py:196-215 for THREE batches on the reference, gates trained by the reference's RAdam as
    its trainer does (the gates are the only student parameters that require grad: create_new_optimizer, :152-162), at every gate
    site class cfg/taylor_importance_track.json uses (conv sites, `bnK.1` ReLUs, `aspp.features.N.2`).  Two variants:
      acc1  accumulation_steps = 1: optimizer.step() + zero_grad() after every batch;
      accN  accumulation_steps = 100000 (the shipped config): step + zero_grad at batch 0 only, so gate.weight.grad keeps
            ACCUMULATING over the later batches and the importance is (gate * running gradient sum)^2.
    Stored per variant and step: loss, gate gradients as read by get_gate_importance, importances, gate values after the
    step; and ImportanceFilterTracker.average() at the end."""
    from models.students.taylor_prune_student import TaylorPruneStudent
    from utils.util import ImportanceFilterTracker
    out = dict(names=np.array([n for n, _ in TAYLOR_STEP_GATES]), lr=np.float64(lr), steps=np.int64(steps))
    xs = [seeded_input(f"taylor.steps.x{i}", (batch, 3) + hw) for i in range(steps)]
    tgts = []
    for i in range(steps):
        t = torch.randint(0, 19, (batch,) + hw, generator=torch.Generator().manual_seed(40 + i))
        t[:, :4] = 255
        tgts.append(t)
        out[f"target{i}"] = t.numpy().astype(np.uint8)
    for variant, acc_steps in (("acc1", 1), ("accN", 100000)):
        teacher = DeepWV3Plus(num_classes=19)
        seeded_fill_(teacher, "teacher.")
        teacher.eval()
        model = TaylorPruneStudent(teacher, config=None)
        model.replace([{"name": n, "epoch": 1, "num_features": c} for n, c in TAYLOR_STEP_GATES])
        params = [p for p in model.student.parameters() if p.requires_grad]
        assert len(params) == len(TAYLOR_STEP_GATES)            # the gates and nothing else
        opt = RAdam(params, lr=lr)
        tr = ImportanceFilterTracker(writer=None)
        tr.update_importance_list(model.added_gates)
        crit = ref_losses.CrossEntropyLoss2d(ignore_index=255)
        for i in range(steps):
            out_st, _ = model(xs[i])
            loss = crit(out_st, tgts[i])
            loss.backward()
            imp = model.get_gate_importance()
            tr.update(imp)
            out[f"{variant}.loss{i}"] = np.float64(loss.item())
            for n, _ in TAYLOR_STEP_GATES:
                out[f"{variant}.grad{i}:{n}"] = model.added_gates[n].weight.grad.detach().numpy().copy()
                out[f"{variant}.imp{i}:{n}"] = np.asarray(imp[n], np.float64)
            if i % acc_steps == 0:
                opt.step()
                opt.zero_grad()
            for n, _ in TAYLOR_STEP_GATES:
                out[f"{variant}.gate{i}:{n}"] = model.added_gates[n].weight.detach().numpy().copy()
        avg = tr.average()
        for n, _ in TAYLOR_STEP_GATES:
            out[f"{variant}.avg:{n}"] = np.asarray(avg[n], np.float64)
    # gates away from 1 (0.5 .. 1.5, seeded): one batch -- pins that a gate scales the forward and that its gradient is taken
    # with respect to the gate, not the gated tensor (the RAdam steps above move the gates by ~1e-5 only)
    teacher = DeepWV3Plus(num_classes=19)
    seeded_fill_(teacher, "teacher.")
    teacher.eval()
    model = TaylorPruneStudent(teacher, config=None)
    model.replace([{"name": n, "epoch": 1, "num_features": c} for n, c in TAYLOR_STEP_GATES])
    with torch.no_grad():
        for n, c in TAYLOR_STEP_GATES:
            model.added_gates[n].weight.copy_(0.5 + seeded_input(f"taylor.gate.{n}", (c,)).sigmoid())
    out_st, _ = model(xs[0])
    loss = ref_losses.CrossEntropyLoss2d(ignore_index=255)(out_st, tgts[0])
    loss.backward()
    imp = model.get_gate_importance()
    out["rnd.loss"] = np.float64(loss.item())
    out.update({f"rnd.logits.{k}": v for k, v in summarize(out_st).items()})
    for n, _ in TAYLOR_STEP_GATES:
        out[f"rnd.gate:{n}"] = model.added_gates[n].weight.detach().numpy().copy()
        out[f"rnd.grad:{n}"] = model.added_gates[n].weight.grad.detach().numpy().copy()
        out[f"rnd.imp:{n}"] = np.asarray(imp[n], np.float64)
    save("taylor_steps", **out)


def g_tta(h=40, w=72, crop=32, classes=19):
    """utils/tta_process.py on arrays (get_crops_image -> reverse_mapping, the numpy half of DepthwiseStudent.inference_test,
    models/students/depthwise_student.py:187-206) for scales = [1.0] on an image that needs 3 x 2 overlapping windows: the
    window boxes and order, the (reference-indexed) window count normalisation, the flip restore and the mean.  The student is
    replaced by seeded per-window logits (stored), so only the reference's tiling / merge arithmetic is pinned.  cv2.resize is
    only ever asked for the identity resize at scale 1.0."""
    import cv2 as cv2_stub
    from utils import tta_process as tta
    np.float = float                                    # tta_process.py:48 uses the alias numpy removed

    def same_size_resize(x, size, interpolation=None):
        assert (x.shape[1], x.shape[0]) == tuple(size), "only the identity resize (scale 1.0) is pinned"
        return x.copy()
    cv2_stub.resize, cv2_stub.INTER_LINEAR = same_size_resize, 1
    tta.cv2 = cv2_stub
    img = seeded_input("tta.img", (3, h, w))
    image_data = ((w, h), [[img, torch.flip(img, dims=[2])]])
    ori_size, mapping, tensors = tta.get_crops_image(image_data, [1.0], crop_size=crop)
    results = seeded_input("tta.logits", (tensors.shape[0], classes, crop, crop)).numpy()
    with np.errstate(divide="ignore", invalid="ignore"):
        out = tta.reverse_mapping(mapping, results, ori_size)
    merged = np.mean(out, axis=0)
    # (the windows and the stand-in logits are seeded: the test regenerates them from their keys)
    save("tta", hw=np.array([h, w]), crop=np.int64(crop), boxes=np.array(mapping[0][2], np.int64), merged=merged.astype(np.float32),
         windows_sum=np.float64(tensors.double().sum().item()))


def _canny_stub_map(shape, seed):
    """The edge map handed to both sides in place of cv2.Canny's output (0 / 255, like cv2): parity of the Canny operator
    itself is unpinned (opencv-python is not vendored); everything downstream of it is pinned by these goldens."""
    g = torch.Generator().manual_seed(seed)
    return ((torch.rand(shape, generator=g) < 0.12).float() * 255.0).numpy()


def g_gscnn(hw=(64, 128), batch=2):
    """models/gscnn/gscnn.py:183-325 GSCNN(19).forward in eval mode on seeded weights, with cv2.Canny replaced by a seeded
    0/255 map (recorded by seed): logits + the sub-graph anchors (edge attention `acts`, ASPP output); plus the building
    blocks alone: GatedSpatialConv2d (gate_spatial_conv.py:17-65) and Resnet.BasicBlock (encoders/Resnet.py:64-99)."""
    import json
    import cv2 as cv2_stub
    from models.gscnn.gscnn import GSCNN
    from models.gscnn import gate_spatial_conv as gsc
    from models.encoders import Resnet
    torch.Tensor.cuda = lambda self, *a, **k: self          # gscnn.py:288 calls .cuda() on the canny tensor
    with torch.device("meta"):
        m = GSCNN(num_classes=19)
    inv = {k: list(v.shape) for k, v in m.state_dict().items()}
    with open(os.path.join(OUT, "gscnn_keys.json"), "w") as f:
        json.dump(inv, f)
    net = GSCNN(num_classes=19)
    seeded_fill_(net, "gscnn.")
    net.eval()
    x = seeded_input("gscnn.x", (batch, 3) + hw, scale=40.0)   # spread so that the uint8 cast is not all 0 / 1
    calls = []

    def fake_canny(img, lo, hi):
        calls.append((img.shape, img.dtype, lo, hi))
        return _canny_stub_map(hw, 500 + len(calls) - 1)
    cv2_stub.Canny = fake_canny
    keep = {}
    net.cw.register_forward_hook(lambda mod, i, o: keep.__setitem__("cw", o.detach()))
    net.aspp.register_forward_hook(lambda mod, i, o: keep.__setitem__("aspp", o.detach()))
    net.gate1.register_forward_hook(lambda mod, i, o: keep.__setitem__("gate1", o.detach()))
    with torch.no_grad():
        y = net(x)
    assert calls == [((hw[0], hw[1], 3), np.dtype("uint8"), 10, 100)] * batch, calls
    out = dict(x_key="gscnn.x", x_scale=np.float64(40.0), canny_seeds=np.array([500 + i for i in range(batch)]),
               logits=summarize(y), acts=summarize(torch.sigmoid(keep["cw"])), aspp=summarize(keep["aspp"]),
               gate1=summarize(keep["gate1"]))
    # building blocks
    g = gsc.GatedSpatialConv2d(16, 16); seeded_fill_(g, "gscnn.blk.gate."); g.eval()
    f, a = seeded_input("gscnn.blk.gate.f", (2, 16, 12, 20)), seeded_input("gscnn.blk.gate.a", (2, 1, 12, 20))
    with torch.no_grad():
        out["blk_gate.y"] = g(f, a).numpy()
    b = Resnet.BasicBlock(16, 16, stride=1, downsample=None); seeded_fill_(b, "gscnn.blk.res."); b.eval()
    xb = seeded_input("gscnn.blk.res.x", (2, 16, 12, 20))
    with torch.no_grad():
        out["blk_res.y"] = b(xb).numpy()
    save("gscnn", **out)


GSCNN_FULL_PLAN = ["mod4.block2.convs.conv2", "aspp.features.1.0"]
GSCNN_FULL_HINTS = ["mod4.block2.convs.conv2", "aspp.features.1.0", "aspp"]


def g_gscnn_step_full(hw=(64, 128), batch=2):
    """Mode B on the reference's Gated-SCNN: DepthwiseStudent(GSCNN) with two cheap-conv blocks, EVERY student parameter
    trainable -- trunk, ASPP incl. the edge branch, decoder and the whole shape stream (dsn3/4/7, res1-3, d1-3, gate1-3, fuse, cw) --
    hints on a block conv, an ASPP branch conv and the `aspp` module (whose gradient re-enters the trunk through the edge branch
    and the side outputs), loss = KLDivergenceLoss(1)(student, teacher logits) + sum of hint MSEs, loss.backward()
    (models/gscnn/gscnn.py:269-325 under autograd).  cv2.Canny is replaced by seeded 0/255 maps on both networks, as in g_gscnn.
    Stores the losses, logits and a summary of every parameter's gradient."""
    import cv2 as cv2_stub
    from models.gscnn.gscnn import GSCNN
    torch.Tensor.cuda = lambda self, *a, **k: self
    teacher = GSCNN(num_classes=19)
    seeded_fill_(teacher, "gscnn.")
    teacher.eval()
    model = DepthwiseStudent(teacher, config=None)
    model.replace([{"name": n, "epoch": 1} for n in GSCNN_FULL_PLAN], kernel_size=9, padding=20, dilation=5)
    for n in GSCNN_FULL_PLAN:
        seeded_fill_(model.get_block(n, model.student), f"student.{n}.")
    model.register_hint_layers(GSCNN_FULL_HINTS)
    for p in model.student.parameters():
        p.requires_grad = True
    calls = []

    def fake_canny(img, lo, hi):      # teacher forward first, then student: image i of either network gets map 900 + i
        calls.append(1)
        return _canny_stub_map(hw, 900 + (len(calls) - 1) % batch)
    cv2_stub.Canny = fake_canny
    x = seeded_input("gscnn.full.x", (batch, 3) + hw, scale=30.0)
    out_st, out_tc = model(x)
    kd = ref_losses.KLDivergenceLoss(1)(out_st, out_tc)
    crit = ref_losses.MSELoss(num_classes=1000)
    hint, per = 0, []
    for s_, t_ in zip(model.student_hidden_outputs, model.teacher_hidden_outputs):
        l = crit(s_, t_); per.append(l.item()); hint = hint + l
    (kd + hint).backward()
    out = dict(plan=np.array(GSCNN_FULL_PLAN), hints=np.array(GSCNN_FULL_HINTS), x_key="gscnn.full.x", x_scale=np.float64(30.0),
               canny_seeds=np.array([900 + i for i in range(batch)]), kd_loss=np.float64(kd.item()), hint_loss=np.float64(hint.item()),
               per_hint=np.array(per), student_logits=summarize(out_st, 1024))
    names = []
    for n, p in model.student.named_parameters():
        if p.grad is None:
            assert n.startswith("dsn1."), n        # dsn1 is declared but never used by GSCNN.forward (gscnn.py:269-314)
            continue
        names.append(n); out[f"grad:{n}"] = summarize(p.grad, 256)
    out["trainable"] = np.array(names)
    save("gscnn_step_full", **out)


def trainer_config(plan, lr, len_epoch, save_dir):
    """A config dict in the reference's JSON schema (cfg/cityscapes/*.json) for a tiny synthetic run."""
    ent = [{"name": n, "epoch": 1} for n in plan]
    return {
        "name": "golden_trainer", "n_gpu": 0,
        "teacher": {"type": "DeepWV3Plus", "args": {"num_classes": 19}},
        "optimizer": {"type": "RAdam", "args": {"lr": lr}},
        "supervised_loss": {"type": "CrossEntropyLoss2d", "args": {"ignore_index": 255}},
        "kd_loss": {"type": "MSELoss", "args": {"reduction": "mean", "num_classes": 1}},
        "hint_loss": {"type": "MSELoss", "args": {"reduction": "mean", "num_classes": 1000}},
        "metrics": [],
        "lr_scheduler": {"type": "MyReduceLROnPlateau", "args": {"mode": "min", "threshold": 0.01, "factor": 0.5, "patience": 0,
                                                               "verbose": True, "min_lr": 1e-05, "threshold_mode": "rel"}},
        "trainer": {"name": "LayerwiseTrainer", "epochs": 1, "save_dir": save_dir, "save_period": 100, "verbosity": 0,
                    "monitor": "off", "accumulation_steps": 1, "log_step": 100, "do_validation_interval": 100,
                    "len_epoch": len_epoch, "tensorboard": False},
        "pruning": {"args": {"dilation": 5, "padding": 20, "kernel_size": 9}, "pruning_plan": ent, "hint": ent, "unfreeze": ent},
        "weight_scheduler": {"alpha": {"value": 0.0001, "anneal_rate": 2, "max": 0}, "beta": {"value": 0.99, "anneal_rate": 0.95, "min": 0.99},
                             "gamma": {"value": 1, "anneal_rate": 1}},
    }


def g_trainer_epoch():
    """LayerwiseTrainer._train_epoch(1) of the reference on seeded synthetic batches: the log dict after
    len_epoch+1 = 3 iterations and the trainable tensors after 3 RAdam steps (trainer/layerwise_trainer.py:203-305)."""
    import tempfile
    from parse_config import ConfigParser
    from trainer import LayerwiseTrainer
    from utils import WeightScheduler
    from utils import optim as ref_optim
    plan = PLANS["g4"]
    cfgd = trainer_config(plan, lr=1e-3, len_epoch=2, save_dir=tempfile.mkdtemp(prefix="kdgold_"))
    config = ConfigParser(cfgd, run_id="g")
    teacher = DeepWV3Plus(num_classes=19); seeded_fill_(teacher, "teacher."); teacher.eval()
    model = DepthwiseStudent(teacher, config)
    # new blocks must be seeded AFTER replace(): wrap it (replace runs inside prepare_train_epoch)
    orig_replace = model.replace
    def replace_and_seed(blocks, **kw):
        orig_replace(blocks, **kw)
        for b in blocks:
            seeded_fill_(model.get_block(b["name"], model.student), f"student.{b['name']}.")
    model.replace = replace_and_seed
    crit = [config.init_obj(k, ref_losses) for k in ("supervised_loss", "kd_loss", "hint_loss")]
    opt = config.init_obj("optimizer", ref_optim, model.student.parameters())
    sched = config.init_obj("lr_scheduler", ref_optim.lr_scheduler, opt)
    batches = []
    for i in range(3):
        x = seeded_input(f"trainer.x{i}", (2, 3, 64, 128))
        t = torch.randint(0, 19, (2, 64, 128), generator=torch.Generator().manual_seed(100 + i)); t[:, :4] = 255
        batches.append((x, t))
    tr = LayerwiseTrainer(model, crit, [], opt, config, [(x, t.clone()) for x, t in batches], None, sched,
                          WeightScheduler(config["weight_scheduler"]))
    log = tr._train_epoch(1)
    out = {"plan": np.array(plan), "lr": 1e-3, "targets": np.stack([t.numpy() for _, t in batches]).astype(np.uint8)}
    for k, v in log.items():
        out["log:" + k] = np.float64(v)
    for n, p in model.student.named_parameters():
        if p.requires_grad:
            out["param:" + n] = summarize(p.data)
    save("trainer_epoch_g4", **out)


def g_classification_epoch():
    """ClassificationTrainer._train_epoch(1) of the reference (BASELINE config 0 shape: ResNet-20 teacher, identical
    student fully unfrozen, loss = KLDiv(T=5), Adam) on seeded weights / batches: log dict after 3 iterations."""
    import tempfile
    from parse_config import ConfigParser
    from trainer import ClassificationTrainer
    from utils import WeightScheduler
    from utils import optim as ref_optim
    import models.cifar_models as ref_cifar
    import models.metric as ref_metric
    cfgd = trainer_config([], lr=0.01, len_epoch=2, save_dir=tempfile.mkdtemp(prefix="kdgold_"))
    cfgd.update(name="golden_cls", teacher={"type": "resnet20", "args": {}}, optimizer={"type": "Adam", "args": {"lr": 0.01}},
                kd_loss={"type": "KLDivergenceLoss", "args": {"temperature": 5}},
                hint_loss={"type": "MSELoss", "args": {"reduction": "mean", "num_classes": 1}},
                metrics=["accuracy", "top_k_acc"],
                lr_scheduler={"type": "MultiStepLR", "args": {"milestones": [15, 25], "gamma": 0.2}})
    cfgd["trainer"]["name"] = "ClassificationTrainer"
    config = ConfigParser(cfgd, run_id="c")
    teacher = ref_cifar.resnet20(); seeded_fill_(teacher, "cifar.teacher."); teacher.eval()
    model = DepthwiseStudent(teacher, config)
    crit = [config.init_obj(k, ref_losses) for k in ("supervised_loss", "kd_loss", "hint_loss")]
    metrics = [getattr(ref_metric, m) for m in config["metrics"]]
    opt = config.init_obj("optimizer", ref_optim, model.student.parameters())
    sched = config.init_obj("lr_scheduler", ref_optim.lr_scheduler, opt)
    batches = [(seeded_input(f"cls.x{i}", (32, 3, 32, 32)),
                torch.randint(0, 10, (32,), generator=torch.Generator().manual_seed(200 + i))) for i in range(3)]
    tr = ClassificationTrainer(model, crit, metrics, opt, config, batches, None, sched, WeightScheduler(config["weight_scheduler"]))
    log = tr._train_epoch(1)
    out = {"targets": np.stack([t.numpy() for _, t in batches])}
    for k, v in log.items():
        out["log:" + k] = np.float64(v)
    out["param:linear.weight"] = model.student.linear.weight.detach().numpy()
    out["param:conv1.weight"] = model.student.conv1.weight.detach().numpy()
    out["buf:bn1.running_mean"] = model.student.bn1.running_mean.numpy()
    save("classification_epoch", **out)


def g_confusion():
    """utils/util.py:73-128 CityscapesMetricTracker: two update() calls on seeded logits / labels (ignore band, exact ties,
    a class that never occurs) -> the accumulated 19x19 confusion matrix and get_iou()."""
    from utils.util import CityscapesMetricTracker
    tr = CityscapesMetricTracker()
    xs, ts = [], []
    for i in range(2):
        x = seeded_input(f"conf.x{i}", (2, 19, 24, 40))
        x[:, 17] = -100.0                       # class 17 is never predicted
        x[0, 3, :4] = x[0, 5, :4]               # exact ties: argmax must take the first maximum
        x[0, 3, :4] += 50.0; x[0, 5, :4] += 50.0
        t = torch.randint(0, 19, (2, 24, 40), generator=torch.Generator().manual_seed(40 + i))
        t[t == 16] = 3                          # class 16 never occurs as a label
        t[:, :3] = 255
        xs.append(x.numpy().copy()); ts.append(t.numpy().astype(np.uint8))
        tr.update(x.clone(), t.clone())         # (update rewrites 255 -> 19 in the tensor it is given)
    save("confusion", x=np.stack(xs), target=np.stack(ts), conf=tr.conf.astype(np.int64), miou=np.float64(tr.get_iou()))


def g_keys():
    """State-dict key / shape inventory of the reference's DeepWV3Plus(19) (the checkpoint contract)."""
    import json
    with torch.device("meta"):
        m = DeepWV3Plus(num_classes=19)
    inv = {k: list(v.shape) for k, v in m.state_dict().items()}
    path = os.path.join(OUT, "deepwv3plus_keys.json")
    with open(path, "w") as f:
        json.dump(inv, f)
    print("wrote", path, len(inv), "entries", sum(int(np.prod(v)) for k, v in inv.items() if "num_batches" not in k and "running" not in k), "params")


ALL = dict(keys=g_keys, gscnn=g_gscnn, confusion=g_confusion, taylor=g_taylor, taylor_steps=g_taylor_steps, tta=g_tta, gscnn_step_full=g_gscnn_step_full, student_step_full=g_student_step_full, trainer_epoch=g_trainer_epoch, classification_epoch=g_classification_epoch, losses=g_losses, variants=g_variants, dwsep=g_dwsep, resblock=g_resblock, aspp=g_aspp, ops=g_ops, radam=g_radam,
           student_step=g_student_step)

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default=None)
    a = ap.parse_args()
    os.makedirs(OUT, exist_ok=True)
    for k, fn in ALL.items():
        if a.only in (None, k):
            fn()
