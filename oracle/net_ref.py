"""Network-level CPU oracle: the reference's KD train step restated with stock torch CPU ops.

TEST INFRASTRUCTURE ONLY (tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg) -- never imported by the
product package and independent of it: the graph is written here functionally over a flat {checkpoint key: tensor}
dict, following the reference line by line:
  DeepWV3Plus.forward                      models/deeplabv3/deeplabv3.py:141-162
  _AtrousSpatialPyramidPoolingModule       models/deeplabv3/deeplabv3.py:64-75
  IdentityResidualBlock.forward            models/encoders/wider_resnet.py:169-182  (incl. the in-place add that
                                           aliases a hook on the block's last conv with the block output)
  WiderResNetA2 layout                     models/encoders/wider_resnet.py:304-356
  DepthwiseSeparableBlock                  models/students/transform_blocks/depthwise_separable_conv.py:4-13
  DepthwiseStudent.forward / hooks         models/students/depthwise_student.py:46-78,168-177
  the four criteria + loss = hint_loss     trainer/layerwise_trainer.py:223-235, losses/*.py
Pinned by tests/golden/student_step_g4.npz (outputs of the reference itself), see tests/test_oracle_net.py.
"""
import torch
import torch.nn.functional as F

CHANNELS = [(128, 128), (256, 256), (512, 512), (512, 1024), (512, 1024, 2048), (1024, 2048, 4096)]
STRUCTURE = [3, 3, 6, 3, 1, 1]
EPS = 1e-5


def _bnrelu(sd, p, x, relu=True):
    """bnrelu Sequential(BatchNorm2d, ReLU) (wider_resnet.py:43-48); `p` names the BN.  A `<relu>.gate` entry is the GateLayer
    TaylorPruneStudent puts behind that ReLU: `<seq>.1.gate` for a block's `bnK` (p = `<seq>.0`), `aspp.features.N.2.gate` for
    an ASPP branch (p = `aspp.features.N.1`)."""
    y = F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"], sd[p + ".weight"], sd[p + ".bias"], False, 0.0, EPS)
    if not relu:
        return y
    y = F.relu(y)
    head, idx = p.rsplit(".", 1)
    gate = sd.get(f"{head}.{int(idx) + 1}.gate") if idx.isdigit() else None
    return y if gate is None else y * gate.view(1, -1, 1, 1)


def _conv(sd, name, x, stride, pad, dil, cheap_geom):
    """Dense conv `name`, or -- when the student's dict holds a replaced block there -- depthwise k x k then 1x1.  A
    `<name>.gate` entry is the GateLayer the reference's TaylorPruneStudent puts behind the conv
    (models/students/taylor_prune_student.py:37-40, transform_blocks/gate.py:11-12)."""
    if name + ".gate" in sd:
        gate = sd[name + ".gate"]
        rest = {k: v for k, v in sd.items() if k != name + ".gate"}
        return _conv(rest, name, x, stride, pad, dil, cheap_geom) * gate.view(1, -1, 1, 1)
    if name + ".weight" in sd:
        return F.conv2d(x, sd[name + ".weight"], None, stride, pad, dil)
    k, p, d = cheap_geom
    wd = sd[name + ".separable_conv.weight"]
    y = F.conv2d(x, wd, sd.get(name + ".separable_conv.bias"), 1, p, d, groups=wd.shape[0])
    return F.conv2d(y, sd[name + ".pointwise_conv.weight"], sd.get(name + ".pointwise_conv.bias"))


def _upsample(x, size):
    return F.interpolate(x, size=size, mode="bilinear", align_corners=True)


def _trunk(sd, x, note, cheap_geom):
    """WiderResNetA2 as DeepWV3Plus / GSCNN use it (wider_resnet.py:304-356): returns (m1, m2, m3, m4, m7)."""
    x = F.conv2d(x, sd["mod1.conv1.weight"], None, 1, 1)
    feats = {"m1": x}
    for mod_id, nblocks in enumerate(STRUCTURE):
        mname = f"mod{mod_id + 2}"
        if mod_id < 2:
            x = F.max_pool2d(x, 3, stride=2, padding=1)
        for b in range(nblocks):
            p = f"{mname}.block{b + 1}"
            ch = CHANNELS[mod_id]
            dil = 2 if mod_id == 3 else (4 if mod_id > 3 else 1)
            stride = 2 if (b == 0 and mod_id == 2) else 1
            a = _bnrelu(sd, p + ".bn1.0", x)
            shortcut = F.conv2d(a, sd[p + ".proj_conv.weight"], None, stride) if (p + ".proj_conv.weight") in sd else x
            if len(ch) == 2:
                c1 = _conv(sd, p + ".convs.conv1", a, stride, dil, dil, cheap_geom)
                note(p + ".convs.conv1", c1)
                out = _conv(sd, p + ".convs.conv2", _bnrelu(sd, p + ".convs.bn2.0", c1), 1, dil, dil, cheap_geom)
                last = "conv2"
            else:
                c1 = _conv(sd, p + ".convs.conv1", a, stride, 0, 1, cheap_geom)
                note(p + ".convs.conv1", c1)
                c2 = _conv(sd, p + ".convs.conv2", _bnrelu(sd, p + ".convs.bn2.0", c1), 1, dil, dil, cheap_geom)
                note(p + ".convs.conv2", c2)
                out = _conv(sd, p + ".convs.conv3", _bnrelu(sd, p + ".convs.bn3.0", c2), 1, 0, 1, cheap_geom)  # dropout: eval
                last = "conv3"
            out = out + shortcut           # reference: out.add_(shortcut) -> hooked tensor == block output
            note(p + ".convs." + last, out)
            note(p + ".convs", out)
            note(p, out)                   # a hook on the block itself sees its return value: the same tensor
            x = out
        feats[f"m{mod_id + 2}"] = x
    return feats


def _aspp_branches(sd, x, note, cheap_geom):
    size = x.shape[2:]
    img = F.adaptive_avg_pool2d(x, 1)
    img = _bnrelu(sd, "aspp.img_conv.1", F.conv2d(img, sd["aspp.img_conv.0.weight"]))
    outs = [_upsample(img, size)]
    feats = []
    for i, r in enumerate([None, 12, 24, 36]):
        n = f"aspp.features.{i}.0"
        y = _conv(sd, n, x, 1, 0 if r is None else r, 1 if r is None else r, cheap_geom)
        note(n, y)
        feats.append(_bnrelu(sd, f"aspp.features.{i}.1", y))
    return outs, feats


def forward(sd, x, hint_names=(), cheap_geom=(9, 20, 5)):
    """Returns (logits, [hint tensors in forward-execution order], [their names])."""
    want = set(hint_names)
    hints, names = [], []

    def note(name, t):
        if name in want:
            hints.append(t)
            names.append(name)

    f = _trunk(sd, x, note, cheap_geom)
    m2, x = f["m2"], f["m7"]
    outs, feats = _aspp_branches(sd, x, note, cheap_geom)
    x = torch.cat(outs + feats, 1)
    note("aspp", x)                        # hook on the ASPP module: the concatenated, activated branches
    dec0_up = F.conv2d(x, sd["bot_aspp.weight"])
    dec0 = torch.cat([F.conv2d(m2, sd["bot_fine.weight"]), _upsample(dec0_up, m2.shape[2:])], 1)
    y = _bnrelu(sd, "final.1", F.conv2d(dec0, sd["final.0.weight"], None, 1, 1))
    y = _bnrelu(sd, "final.4", F.conv2d(y, sd["final.3.weight"], None, 1, 1))
    y = F.conv2d(y, sd["final.6.weight"])
    return _upsample(y, (y.shape[2] * 2, y.shape[3] * 2)), hints, names


# ---- Gated-SCNN (models/gscnn/gscnn.py:183-325): the same trunk plus a full-resolution shape stream -----------------------
def _bn_plain(sd, p, x):
    return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"], sd[p + ".weight"], sd[p + ".bias"], False, 0.0, EPS)


def _basic_block(sd, p, x):
    """Resnet.BasicBlock, stride 1, no downsample (models/encoders/Resnet.py:64-99)."""
    out = F.relu(_bn_plain(sd, p + ".bn1", F.conv2d(x, sd[p + ".conv1.weight"], None, 1, 1)))
    out = _bn_plain(sd, p + ".bn2", F.conv2d(out, sd[p + ".conv2.weight"], None, 1, 1))
    return F.relu(out + x)


def _gated_conv(sd, p, feat, gate):
    """GatedSpatialConv2d.forward (models/gscnn/gate_spatial_conv.py:50-60)."""
    z = _bn_plain(sd, p + "._gate_conv.0", torch.cat([feat, gate], 1))
    z = F.relu(F.conv2d(z, sd[p + "._gate_conv.1.weight"], sd[p + "._gate_conv.1.bias"]))
    a = torch.sigmoid(_bn_plain(sd, p + "._gate_conv.4", F.conv2d(z, sd[p + "._gate_conv.3.weight"], sd[p + "._gate_conv.3.bias"])))
    return F.conv2d(feat * (a + 1), sd[p + ".weight"])


def gscnn_forward(sd, x, canny, hint_names=(), cheap_geom=(9, 20, 5), want_aux=False):
    """GSCNN.forward with the cv2.Canny output given as `canny` (N,1,H,W, values 0/255: gscnn.py:284-288 computes it on the
    host from the uint8-cast input).  Returns (logits, hints, names[, aux])."""
    want = set(hint_names)
    hints, names = [], []

    def note(name, t):
        if name in want:
            hints.append(t)
            names.append(name)

    size = x.shape[2:]
    f = _trunk(sd, x, note, cheap_geom)
    up = lambda t: _upsample(t, size)
    s3 = up(F.conv2d(f["m3"], sd["dsn3.weight"], sd["dsn3.bias"]))
    s4 = up(F.conv2d(f["m4"], sd["dsn4.weight"], sd["dsn4.bias"]))
    s7 = up(F.conv2d(f["m7"], sd["dsn7.weight"], sd["dsn7.bias"]))
    cs = up(_basic_block(sd, "res1", up(f["m1"])))
    cs = _gated_conv(sd, "gate1", F.conv2d(cs, sd["d1.weight"], sd["d1.bias"]), s3)
    g1 = cs
    cs = up(_basic_block(sd, "res2", cs))
    cs = _gated_conv(sd, "gate2", F.conv2d(cs, sd["d2.weight"], sd["d2.bias"]), s4)
    cs = up(_basic_block(sd, "res3", cs))
    cs = _gated_conv(sd, "gate3", F.conv2d(cs, sd["d3.weight"], sd["d3.bias"]), s7)
    edge_out = torch.sigmoid(up(F.conv2d(cs, sd["fuse.weight"])))
    acts = torch.sigmoid(F.conv2d(torch.cat((edge_out, canny), 1), sd["cw.weight"]))
    # edge-aware ASPP (gscnn.py:160-181): [image pooling, edge, 1x1, rates 12/24/36]
    m7 = f["m7"]
    outs, feats = _aspp_branches(sd, m7, note, cheap_geom)
    edge = _bnrelu(sd, "aspp.edge_conv.1", F.conv2d(_upsample(acts, m7.shape[2:]), sd["aspp.edge_conv.0.weight"]))
    xa = torch.cat(outs + [edge] + feats, 1)
    note("aspp", xa)
    dec0_up = F.conv2d(xa, sd["bot_aspp.weight"])
    m2 = f["m2"]
    dec0 = torch.cat([F.conv2d(m2, sd["bot_fine.weight"]), _upsample(dec0_up, m2.shape[2:])], 1)
    y = _bnrelu(sd, "final_seg.1", F.conv2d(dec0, sd["final_seg.0.weight"], None, 1, 1))
    y = _bnrelu(sd, "final_seg.4", F.conv2d(y, sd["final_seg.3.weight"], None, 1, 1))
    y = F.conv2d(y, sd["final_seg.6.weight"])
    logits = F.interpolate(y, size=size, mode="bilinear", align_corners=False)      # gscnn.py:323: no align_corners
    if want_aux:
        return logits, hints, names, dict(acts=acts, aspp=xa, gate1=g1)
    return logits, hints, names


def kl_div_loss(s, t, T=1.0):  # losses/KLDiv.py:19-23
    return F.kl_div(F.log_softmax(s / T, dim=1), F.softmax(t / T, dim=1), reduction="mean") * (T ** 2) * t.shape[1]


def mse_loss(s, t, num_classes):  # losses/MSELoss.py:14-16
    return F.mse_loss(s, t) * num_classes


def weighted_hint_loss(s, t, w):  # losses/WeightedHintMSELoss.py:12-16
    m = ((s - t) ** 2).mean(dim=(-1, -2))
    return ((w * m).sum(dim=-1) / w.sum(dim=-1)).mean()


def kd_step(teacher_sd, student_sd, x, target, plan, hint_num_classes=1000, temperature=1.0, cheap_geom=(9, 20, 5),
            hint_weights=None, backprop="hint", hint_names=None, canny=None):
    """One step.  backprop="hint": the reference-faithful loss = hint loss only (trainer/layerwise_trainer.py:233-235);
    backprop="kd+hint": loss = KLDiv(student, teacher logits) + hint loss (SURVEY 8d mode B; the KD term is what
    trainer/classification_trainer.py:37 back-propagates).  student_sd tensors with requires_grad=True are the trainable set
    (with every tensor trainable this includes the eval-mode BN weights / biases, like the reference's "identical
    architecture" branch, layerwise_trainer.py:88-100); returns a dict of losses / outputs / gradients."""
    hn = plan if hint_names is None else hint_names
    fwd = forward if canny is None else (lambda sd, xx, h, cg: gscnn_forward(sd, xx, canny, h, cg))   # canny given: GSCNN
    with torch.no_grad():
        t_logits, t_hints, _ = fwd(teacher_sd, x, hn, cheap_geom)
    s_logits, s_hints, names = fwd(student_sd, x, hn, cheap_geom)
    hint = 0
    per = []
    for i, (s, t) in enumerate(zip(s_hints, t_hints)):
        l = mse_loss(s, t, hint_num_classes) if hint_weights is None else weighted_hint_loss(s, t, hint_weights[i])
        per.append(l)
        hint = hint + l
    train = {k: v for k, v in student_sd.items() if v.requires_grad}
    loss = hint
    if backprop == "kd+hint":
        loss = kl_div_loss(s_logits, t_logits, temperature) + hint
    elif backprop != "hint":
        raise ValueError(backprop)
    grads = torch.autograd.grad(loss, list(train.values()), allow_unused=True) if train else []
    hint = hint.detach() if torch.is_tensor(hint) else torch.tensor(float(hint))
    out = dict(hint_loss=hint, loss=loss.detach(), per_hint=[p.detach() for p in per], student_logits=s_logits.detach(),
               teacher_logits=t_logits, student_hints=[h.detach() for h in s_hints], teacher_hints=t_hints,
               hint_names=names, grads=dict(zip(train.keys(), grads)),
               kd_loss=kl_div_loss(s_logits.detach(), t_logits, temperature),
               kd_mse=mse_loss(s_logits.detach(), t_logits, 1))
    if target is not None:
        out["supervised_loss"] = F.cross_entropy(s_logits.detach(), target, ignore_index=255)
        out["teacher_loss"] = F.cross_entropy(t_logits, target, ignore_index=255)
    return out


def make_student_sd(teacher_sd, plan, new_weights, trainable=None):
    """Student dict = teacher dict with each plan entry's dense weight replaced by the cheap-conv pair."""
    sd = {k: v.detach().clone() for k, v in teacher_sd.items()}
    for name in plan:
        del sd[name + ".weight"]
        for suffix in ("separable_conv.weight", "pointwise_conv.weight"):
            sd[f"{name}.{suffix}"] = new_weights[f"{name}.{suffix}"].detach().clone()
    if trainable == "all":      # every parameter (not the BN running statistics)
        for k, v in sd.items():
            if not (k.endswith("running_mean") or k.endswith("running_var") or k.endswith("num_batches_tracked")):
                v.requires_grad_(True)
        return sd
    for name in (plan if trainable is None else trainable):
        for suffix in ("separable_conv.weight", "pointwise_conv.weight"):
            sd[f"{name}.{suffix}"].requires_grad_(True)
    return sd


def taylor_importance(teacher_sd, x, target, gates, values=None):
    """trainer/taylor_prune_trainer.py:196-211 for one step: gates (unit, or `values[name]`) behind the named blocks of a copy
    of the teacher -- convs, bnrelu ReLUs (`...bnK.1`), ASPP branch convs / ReLUs (`aspp.features.N.0|2`) -- loss = cross entropy
    of the student logits, importance = (gate * d loss / d gate)^2.  gates: {block name: channels}.
    Returns (loss, {name: gate gradient}, {name: importance})."""
    sd = {k: v.detach().clone() for k, v in teacher_sd.items()}
    gs = {}
    for name, c in gates.items():
        gs[name] = (torch.ones(c) if values is None else values[name].detach().clone().float()).requires_grad_(True)
        sd[name + ".gate"] = gs[name]
    logits, _, _ = forward(sd, x)
    loss = F.cross_entropy(logits, target, ignore_index=255)
    grads = torch.autograd.grad(loss, list(gs.values()))
    gg = dict(zip(gs.keys(), grads))
    return loss.detach(), gg, {n: (gs[n].detach() * g) ** 2 for n, g in gg.items()}
