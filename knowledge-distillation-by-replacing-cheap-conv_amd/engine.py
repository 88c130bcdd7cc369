"""Student execution engine: runs a DeepWV3Plus student (with any set of its 3x3 convs replaced by
DepthwiseSeparableBlocks) as a static sequence of HIP kernel launches, forward and backward.

What it replaces in the reference: `self.student(x)` inside DepthwiseStudent.forward
(models/students/depthwise_student.py:168-177), i.e. DeepWV3Plus.forward
(models/deeplabv3/deeplabv3.py:141-162) over IdentityResidualBlock.forward
(models/encoders/wider_resnet.py:169-182) and the ASPP module (deeplabv3.py:64-75), together with
the forward hooks that collect hint features (depthwise_student.py:63-76) and the autograd backward
that loss.backward() (trainer/layerwise_trainer.py:235) runs through that graph.

Design (MI355X-first, not a module-by-module translation):
  * activations live in NHWC, bf16 (or fp32 in parity mode); eval-mode BatchNorm + ReLU never run
    as kernels: each conv's epilogue emits the *activated* input of its consumer (folded
    scale/shift of the consumer's BN) and, only where something needs it, the raw tensor
    (identity shortcuts, hint features, trunk outputs);
  * the in-place residual add is the conv epilogue's `res_pre`; concatenations are channel-slice
    writes into one wide buffer (ASPP 1280 ch, decoder 304 -> 320 ch);
  * backward is hand-scheduled: only the sub-graph between the first trainable block and the
    last hint is visited; d(ReLU o BN)/dx is the dgrad kernels' mask epilogue; frozen dense convs
    get no wgrad, trainable ones are the depthwise + pointwise pair;
  * hint features captured by name at plan time (no Python hooks), reproducing the reference's
    aliasing: a hint on the last conv of a residual block is the block output.
Parameters are read from the nn.Module tree (fp32 masters, reference checkpoint keys); packed
operands are cached per parameter version, so frozen weights are packed once.
"""
import torch
from torch import nn

from . import ops
from ._lib import KD_PACK_DGRAD, KD_PACK_FWD
from .models.students.transform_blocks import DepthwiseSeparableBlock
from .models.wider_resnet import IdentityResidualBlock


class EngineError(RuntimeError):
    pass


def _is_trainable(mod):
    return any(p.requires_grad for p in mod.parameters())


class _Site:
    """One convolution site of a residual block / ASPP branch: dense nn.Conv2d or a cheap-conv block."""

    def __init__(self, name, mod):
        self.name, self.mod = name, mod
        self.cheap = isinstance(mod, DepthwiseSeparableBlock)
        if not self.cheap and not isinstance(mod, nn.Conv2d):
            raise EngineError(f"{name}: unsupported module {type(mod).__name__} in the student graph")
        conv = mod.pointwise_conv if self.cheap else mod
        self.cout = conv.out_channels
        if self.cheap:
            self.k, self.pad, self.dil = mod.geometry
            self.stride = 1
            if mod.separable_conv.bias is not None or mod.pointwise_conv.bias is not None:
                raise EngineError(f"{name}: biased cheap conv inside the fused graph is not supported")
        else:
            if conv.bias is not None or conv.groups != 1:
                raise EngineError(f"{name}: only bias-free dense convs are supported")
            self.k, self.pad, self.dil, self.stride = conv.kernel_size[0], conv.padding[0], conv.dilation[0], conv.stride[0]
        self.trainable = _is_trainable(mod)
        if self.trainable and not self.cheap:
            raise EngineError(f"{name}: weight gradients of dense convs are not implemented (only cheap-conv blocks train; "
                              "reference-faithful mode, SURVEY F4)")


class StudentEngine:
    def __init__(self, net, dtype=torch.bfloat16):
        self.net = net
        self.dtype = dtype
        self.hint_names = []
        self._pack = {}   # (id(param), version, tag) -> packed tensor
        self._bn = {}     # id(bn) -> (scale, shift)
        self._tape = None
        self.last_hint_names = []
        self.reducer = None   # optional parallel.GradReducer: gradients are written into its buckets and announced

    def _grad_like(self, p):
        if self.reducer is not None:
            return self.reducer.grad_buffer(p)
        return torch.empty_like(p, dtype=torch.float32)

    def _grad_done(self, p):
        if self.reducer is not None:
            self.reducer.grad_ready(p)

    # ------------------------------------------------------------------ parameter operands
    def _packed(self, p, tag, fn):
        # keyed by id(p), but the entry holds p itself: a Parameter freed by a later replace() cannot hand its id (and a
        # matching small _version) to a new one while the entry is alive, and `ent[2] is p` catches any other aliasing
        key = (id(p), tag)
        ent = self._pack.get(key)
        if ent is None or ent[2] is not p or ent[0] != p._version or ent[1].device != p.device:
            ent = (p._version, fn(), p)
            self._pack[key] = ent
        return ent[1]

    def drop_caches(self):
        """Forget packed weights / folded BN vectors (called when the module tree is edited: replace(), reset())."""
        self._pack.clear()
        self._bn.clear()

    def _w_fwd(self, conv, cin_pad=None):
        return self._packed(conv.weight, ("fwd", self.dtype, cin_pad),
                            lambda: ops.pack_conv_weight(conv.weight, self.dtype, KD_PACK_FWD, cin_pad))

    def _w_dgrad(self, conv):
        return self._packed(conv.weight, ("dgrad", self.dtype), lambda: ops.pack_conv_weight(conv.weight, self.dtype, KD_PACK_DGRAD))

    def _w_dw(self, conv, flip):
        return self._packed(conv.weight, ("dw", flip), lambda: ops.pack_dw_weight(conv.weight, flip))

    def _bn_fold(self, bn_seq):
        """bnrelu Sequential(BatchNorm2d, ReLU) or a bare BatchNorm2d -> cached (scale, shift)."""
        bn = bn_seq[0] if isinstance(bn_seq, nn.Sequential) else bn_seq
        if bn.training:
            raise EngineError("the fused student graph implements eval-mode BatchNorm only (LayerwiseTrainer keeps the "
                              "student in eval mode, SURVEY F3)")
        if bn.weight.requires_grad or bn.bias.requires_grad:
            raise EngineError("trainable BatchNorm parameters are not supported in the fused graph")
        key = id(bn)
        ver = (bn.weight._version, bn.bias._version, bn.running_mean._version, bn.running_var._version, bn.weight.device)
        ent = self._bn.get(key)
        if ent is None or ent[2] is not bn or ent[0] != ver:
            ent = (ver, ops.bn_fold(bn), bn)
            self._bn[key] = ent
        return ent[1]

    def _new(self, N, H, W, C, dtype=None, zero=False):
        f = torch.zeros if zero else torch.empty
        return f((N, H, W, C), dtype=dtype or self.dtype, device=self.device)

    # ------------------------------------------------------------------ forward
    def forward(self, x, collect_hints=True):
        """x: (N,3,H,W) fp32.  Returns (logits (N,H,W,19) fp32 NHWC, [hint tensors (N,h,w,C) NHWC] in forward order)."""
        net = self.net
        if x.dim() != 4 or x.shape[1] != 3:
            raise EngineError(f"expected an (N,3,H,W) batch, got {tuple(x.shape)}")
        if x.shape[2] % 8 or x.shape[3] % 8:
            raise EngineError("input height/width must be multiples of 8 (two stride-2 pools and one stride-2 conv)")
        self.device = x.device
        N, _, H, W = x.shape
        want = set(self.hint_names) if collect_hints else set()
        seen = []
        hints = []
        tape = {"blocks": [], "aspp": None, "hint_slots": []}

        def note_hint(name, tensor, slot):
            # slot: ("block", block_index, site_index | "out") or ("aspp", branch)
            seen.append(name)
            hints.append(tensor)
            tape["hint_slots"].append(slot)

        xin = x.detach()
        if xin.dtype != torch.float32 or not xin.is_contiguous():
            xin = xin.float().contiguous()
        stem_w = net.mod1.conv1.weight.detach()
        s = ops.stem_conv(xin, stem_w if stem_w.is_contiguous() else stem_w.contiguous(), self.dtype)

        # trunk: (module name, list of blocks)
        mods = [(f"mod{i}", getattr(net, f"mod{i}")) for i in range(2, 8)]
        flat = [(f"{mn}.{bn}", blk) for mn, m in mods for bn, blk in m.named_children()]
        for name, blk in flat:
            if not isinstance(blk, IdentityResidualBlock):
                raise EngineError(f"{name}: expected an IdentityResidualBlock")

        # pool2 (+ bn1 of mod2.block1)
        sc1, sh1 = self._bn_fold(flat[0][1].bn1)
        _, a = ops.maxpool3x3s2(s, sc1, sh1, want_raw=False)
        x_raw, rg = None, False
        m2 = None
        for bi, (name, blk) in enumerate(flat):
            last_of_mod2 = name.startswith("mod2.") and (bi + 1 == len(flat) or not flat[bi + 1][0].startswith("mod2."))
            is_last = bi + 1 == len(flat)
            nxt = None if (is_last or last_of_mod2) else flat[bi + 1][1]
            need_raw = is_last or last_of_mod2 or (nxt is not None and not hasattr(nxt, "proj_conv"))
            x_raw, a, rg, rec = self._block_fwd(name, blk, x_raw, a, rg, nxt.bn1 if nxt is not None else None, need_raw,
                                                want, note_hint, bi)
            tape["blocks"].append(rec)
            if last_of_mod2:
                m2 = x_raw
                sc, sh = self._bn_fold(flat[bi + 1][1].bn1)
                _, a = ops.maxpool3x3s2(m2, sc, sh, want_raw=False)  # pool3 + bn1 of mod3.block1
                x_raw = None
                if rg:
                    raise EngineError("trainable blocks in mod2 would need a max-pool backward (not implemented)")
        x7, rg7 = x_raw, rg

        # ASPP -> 1280-channel buffer, branches write their slices
        aspp = net.aspp
        h8, w8 = x7.shape[1], x7.shape[2]
        cat = self._new(N, h8, w8, 1280)
        sc, sh = self._bn_fold(aspp.img_conv[1])
        ops.aspp_image_pool(x7, aspp.img_conv[0].weight.detach(), sc, sh, cat[..., 0:256])
        arec = {"x7": x7, "rg7": rg7, "branches": []}
        for i, br in enumerate(aspp.features):
            site = _Site(f"aspp.features.{i}.0", br[0])
            sc, sh = self._bn_fold(br[1])
            hinted = site.name in want
            raw = self._new(N, h8, w8, 256) if hinted else None
            out = cat[..., 256 * (i + 1):256 * (i + 2)]
            mid = None
            if site.cheap:
                mid = ops.dwconv(x7, self._w_dw(site.mod.separable_conv, False), site.k, site.pad, site.dil)
                ops.conv2d(mid, self._w_fwd(site.mod.pointwise_conv), out_raw=raw, out_act=out, act_scale=sc, act_shift=sh,
                           act_relu=True)
            else:
                ops.conv2d(x7, self._w_fwd(site.mod), 1, site.pad, site.dil, out_raw=raw, out_act=out, act_scale=sc,
                           act_shift=sh, act_relu=True)
            if hinted:
                note_hint(site.name, raw, ("aspp", i))
            arec["branches"].append({"site": site, "mid": mid})
        tape["aspp"] = arec

        # decoder (no gradient reaches it in the reference-faithful mode: loss = hint loss only)
        up_small = self._new(N, h8, w8, 256)
        ops.conv2d(cat, self._w_fwd(net.bot_aspp), out_raw=up_small)
        h2, w2 = m2.shape[1], m2.shape[2]
        dec0 = self._new(N, h2, w2, 320)  # 48 + 256 = 304 channels, zero-padded to the GEMM K granule
        dec0[..., 304:320].zero_()       # (only the 16 pad channels: the slices below fill the rest)
        ops.conv2d(m2, self._w_fwd(net.bot_fine), out_raw=dec0[..., 0:48])
        ops.upsample_bilinear_ac(up_small, (h2, w2), out=dec0[..., 48:304])
        f = net.final
        sc, sh = self._bn_fold(f[1])
        d1 = self._new(N, h2, w2, 256)
        ops.conv2d(dec0, self._w_fwd(f[0], cin_pad=320), 1, 1, 1, out_act=d1, act_scale=sc, act_shift=sh, act_relu=True,
                   algo_cin=304)
        sc, sh = self._bn_fold(f[4])
        d2 = self._new(N, h2, w2, 256)
        ops.conv2d(d1, self._w_fwd(f[3]), 1, 1, 1, out_act=d2, act_scale=sc, act_shift=sh, act_relu=True)
        ncls = f[6].out_channels
        d3 = self._new(N, h2, w2, ncls, dtype=torch.float32)
        ops.conv2d(d2, self._w_fwd(f[6]), out_raw=d3)
        logits = ops.upsample_bilinear_ac(d3, (H, W), out_dtype=torch.float32)

        missing = want - set(seen)
        if missing:
            raise EngineError(f"hint layers not found in the student graph: {sorted(missing)}")
        self._tape = tape
        self.last_hint_names = list(seen)   # forward-execution order (what hooks would have produced)
        return logits, hints

    def _block_fwd(self, name, blk, x_raw, a1, rg_in, next_bn, need_raw, want, note_hint, bi):
        """One pre-activation residual block.  a1 = relu(bn1(x)) already produced by the previous kernel."""
        convs = [(n, m) for n, m in blk.convs.named_children() if n.startswith("conv")]
        bns = {n: m for n, m in blk.convs.named_children() if n.startswith("bn")}
        for n, m in blk.convs.named_children():
            if isinstance(m, (nn.Dropout, nn.Dropout2d)) and m.training:
                raise EngineError(f"{name}: training-mode dropout is not supported (student runs in eval mode, SURVEY F3)")
        sites = [_Site(f"{name}.convs.{n}", m) for n, m in convs]
        N = a1.shape[0]
        rec = {"name": name, "blk": blk, "sites": sites, "a_in": [], "mid": [], "rg_a": [], "hint_raw": [None] * len(sites),
               "x_raw": x_raw, "rg_in": rg_in, "proj": hasattr(blk, "proj_conv")}
        # shortcut
        if rec["proj"]:
            pc = blk.proj_conv
            if _is_trainable(pc):
                raise EngineError(f"{name}.proj_conv: trainable projection convs are not supported")
            ho = ops.conv_out_size(a1.shape[1], 1, pc.stride[0], 0, 1)
            wo = ops.conv_out_size(a1.shape[2], 1, pc.stride[0], 0, 1)
            shortcut = self._new(N, ho, wo, pc.out_channels)
            ops.conv2d(a1, self._w_fwd(pc), pc.stride[0], 0, 1, out_raw=shortcut)
        else:
            if x_raw is None:
                raise EngineError(f"{name}: identity shortcut needs the raw block input")
            shortcut = x_raw
        a, rg = a1, rg_in
        x_out = a_next = None
        for i, site in enumerate(sites):
            last = i + 1 == len(sites)
            rec["a_in"].append(a)
            rec["rg_a"].append(rg)
            ho = ops.conv_out_size(a.shape[1], site.k, site.stride, site.pad, site.dil)
            wo = ops.conv_out_size(a.shape[2], site.k, site.stride, site.pad, site.dil)
            # hint names: the conv itself, or -- for the last conv -- the `convs` Sequential or the whole block (a forward
            # hook on either observes the tensor the in-place add turns into the block output; cfg/cityscapes/
            # 51M_deeplab_incremental.json uses 'mod4.block2.convs' and 'mod7.block1')
            hinted = site.name in want or (last and (f"{name}.convs" in want or name in want))
            kw = {}
            if last:
                want_raw = need_raw or hinted
                raw = self._new(N, ho, wo, site.cout) if want_raw else None
                kw["res_pre"] = shortcut
                if next_bn is not None:
                    sc, sh = self._bn_fold(next_bn)
                    a_next = self._new(N, ho, wo, site.cout)
                    kw.update(out_act=a_next, act_scale=sc, act_shift=sh, act_relu=True)
                kw["out_raw"] = raw
                x_out = raw
            else:
                sc, sh = self._bn_fold(bns[f"bn{i + 2}"])
                act = self._new(N, ho, wo, site.cout)
                raw = self._new(N, ho, wo, site.cout) if hinted else None
                kw.update(out_raw=raw, out_act=act, act_scale=sc, act_shift=sh, act_relu=True)
            if site.cheap:
                mid = ops.dwconv(a, self._w_dw(site.mod.separable_conv, False), site.k, site.pad, site.dil)
                ops.conv2d(mid, self._w_fwd(site.mod.pointwise_conv), **kw)
            else:
                mid = None
                ops.conv2d(a, self._w_fwd(site.mod), site.stride, site.pad, site.dil, **kw)
            rec["mid"].append(mid)
            rg = rg or site.trainable
            if hinted:
                # forward hooks fire in execution order; the last conv's hooked tensor is mutated by the in-place
                # residual add, so the hint IS the block output
                if last:
                    for nm in (site.name, f"{name}.convs", name):
                        if nm in want:
                            note_hint(nm, raw, ("block", bi, "out"))
                else:
                    rec["hint_raw"][i] = raw
                    note_hint(site.name, raw, ("block", bi, i))
            if not last:
                a = act
        rec["rg_out"] = rg  # rg already includes rg_in (shortcut path)
        return x_out, a_next, rg, rec

    # ------------------------------------------------------------------ backward
    def grad_production_order(self):
        """Trainable parameters in the order backward produces their gradients (ASPP branches first, then the trunk
        blocks from mod7 back to the first trainable block; within a block last conv first, pointwise before depthwise)."""
        net, order = self.net, []

        def add(mod):
            if isinstance(mod, DepthwiseSeparableBlock):
                for p in (mod.pointwise_conv.weight, mod.separable_conv.weight):
                    if p.requires_grad:
                        order.append(p)
        for br in net.aspp.features:
            add(br[0])
        blocks = [blk for i in range(2, 8) for _, blk in getattr(net, f"mod{i}").named_children()]
        for blk in reversed(blocks):
            for _, m in reversed([(n, m) for n, m in blk.convs.named_children() if n.startswith("conv")]):
                add(m)
        return order

    def backward(self, hint_grads):
        """hint_grads: list aligned with forward()'s hints; entries are (N,h,w,C) NHWC tensors or None.
        Returns {parameter: fp32 gradient} for every trainable parameter reached."""
        tape = self._tape
        if tape is None:
            raise EngineError("backward() called without a recorded forward()")
        if len(hint_grads) != len(tape["hint_slots"]):
            raise EngineError("hint gradient list does not match the recorded hints")
        grads = {}
        g_block_out = {}   # block index -> grad wrt raw block output
        g_site_hint = {}   # (block index, site index) -> hint grad of that site's raw output
        g_aspp = {}
        for g, slot in zip(hint_grads, tape["hint_slots"]):
            if g is None:
                continue
            g = self._as_nhwc(g)
            if slot[0] == "aspp":
                g_aspp[slot[1]] = g if slot[1] not in g_aspp else g_aspp[slot[1]] + g
            elif slot[2] == "out":
                g_block_out[slot[1]] = g if slot[1] not in g_block_out else g_block_out[slot[1]] + g
            else:
                key = (slot[1], slot[2])
                g_site_hint[key] = g if key not in g_site_hint else g_site_hint[key] + g

        # ASPP branches: the only consumers of mod7's output that carry gradient (decoder is outside the loss)
        arec = tape["aspp"]
        g_x7 = None
        for i, br in enumerate(arec["branches"]):
            g = g_aspp.get(i)
            if g is None:
                continue
            site = br["site"]
            if site.cheap:
                g_in = self._cheap_bwd(site, arec["x7"], br["mid"], g, grads, need_in=arec["rg7"], res_post=g_x7)
            else:
                g_in = self._dense_dgrad(site, g, res_post=g_x7) if arec["rg7"] else None
            if g_in is not None:
                g_x7 = g_in
        nb = len(tape["blocks"])
        if g_x7 is not None:
            g_block_out[nb - 1] = g_x7 if (nb - 1) not in g_block_out else g_block_out[nb - 1].add_(g_x7)

        for bi in range(nb - 1, -1, -1):
            rec = tape["blocks"][bi]
            g_out = g_block_out.pop(bi, None)
            has_inner = any((bi, i) in g_site_hint for i in range(len(rec["sites"])))
            if g_out is None and not has_inner:
                continue
            g_xin = self._block_bwd(bi, rec, g_out, g_site_hint, grads)
            if g_xin is not None:
                if bi == 0:
                    raise EngineError("gradient reached the stem: not supported")
                prev = tape["blocks"][bi - 1]
                if prev["name"].split(".")[0] != rec["name"].split(".")[0] and prev["name"].startswith("mod2."):
                    raise EngineError("gradient through pool3 is not supported")
                g_block_out[bi - 1] = g_xin if (bi - 1) not in g_block_out else g_block_out[bi - 1].add_(g_xin)
        self._tape = None
        if self.reducer is not None:
            self.reducer.finish()   # current stream waits for the (already overlapped) bucket all-reduces
        return grads

    def _as_nhwc(self, g):
        """Accept an NCHW-logical gradient (the autograd view) or an NHWC tensor; return dense NHWC in the engine dtype."""
        if g.dim() != 4:
            raise EngineError("hint gradients must be 4-D")
        if g.stride(1) == 1 and g.stride(3) != 1:  # logical NCHW over NHWC memory
            g = g.permute(0, 2, 3, 1)
        elif g.stride(3) != 1:
            g = g.permute(0, 2, 3, 1).contiguous()
        if g.dtype != self.dtype:
            g = g.to(self.dtype)
        if not g.is_contiguous():
            g = g.contiguous()
        return g

    def _dense_dgrad(self, site, g, **ep):
        if site.stride != 1:
            raise EngineError(f"{site.name}: input gradient of a strided conv is not implemented")
        N, H, W, _ = g.shape
        conv = site.mod
        out = self._new(N, H, W, conv.in_channels)
        ops.conv2d(g, self._w_dgrad(conv), 1, site.dil * (site.k - 1) - site.pad, site.dil, out_raw=out, **ep)
        return out

    def _cheap_bwd(self, site, a_in, mid, g, grads, need_in, **ep):
        """Backward of dw -> pw: both weight gradients, and (optionally) the input gradient with epilogue `ep`."""
        dw, pw = site.mod.separable_conv, site.mod.pointwise_conv
        if pw.weight.requires_grad:
            gw = self._grad_like(pw.weight)
            ops.pw_wgrad(mid, g, gw)
            grads[pw.weight] = gw
            self._grad_done(pw.weight)
        if not (dw.weight.requires_grad or need_in):
            return None
        N, H, W, _ = g.shape
        g_mid = self._new(N, H, W, pw.in_channels)
        ops.conv2d(g, self._w_dgrad(pw), out_raw=g_mid)
        if dw.weight.requires_grad:
            gw = self._grad_like(dw.weight)
            ops.dwconv_wgrad(a_in, g_mid, gw, site.k, site.pad, site.dil)
            grads[dw.weight] = gw
            self._grad_done(dw.weight)
        if not need_in:
            return None
        return ops.dwconv(g_mid, self._w_dw(dw, True), site.k, site.dil * (site.k - 1) - site.pad, site.dil, **ep)

    def _block_bwd(self, bi, rec, g_out, g_site_hint, grads):
        blk, sites = rec["blk"], rec["sites"]
        bns = {n: m for n, m in blk.convs.named_children() if n.startswith("bn")}
        g = g_out  # gradient w.r.t. the raw output of the current site (last site: the block output)
        for i in range(len(sites) - 1, -1, -1):
            site, a_in, need_in = sites[i], rec["a_in"][i], rec["rg_a"][i]
            g_in = None
            if g is not None and (need_in or site.trainable):
                if i > 0:
                    # a_in = relu(bn_{i+1}(c_{i-1})): mask epilogue gives d/dc_{i-1}; add that tensor's own hint gradient
                    sc, _ = self._bn_fold(bns[f"bn{i + 1}"])
                    ep = dict(mask=a_in, mask_scale=sc, res_post=g_site_hint.get((bi, i - 1)))
                else:
                    sc, _ = self._bn_fold(blk.bn1)
                    ep = dict(mask=a_in, mask_scale=sc)
                    if need_in and g_out is not None:
                        if rec["proj"]:
                            pc = blk.proj_conv
                            psite = _Site(f"{rec['name']}.proj_conv", pc)
                            ep["res_pre"] = self._dense_dgrad(psite, g_out)
                        else:
                            ep["res_post"] = g_out
                if site.cheap:
                    g_in = self._cheap_bwd(site, a_in, rec["mid"][i], g, grads, need_in, **ep)
                elif need_in:
                    g_in = self._dense_dgrad(site, g, **ep)
            elif i > 0 and need_in and g_site_hint.get((bi, i - 1)) is not None:
                pass  # handled below: only the hint gradient flows
            if i > 0:
                if g_in is None:
                    g_in = g_site_hint.get((bi, i - 1)) if rec["rg_a"][i] else None
                g = g_in
            else:
                return g_in if rec["rg_in"] else None
        return None


class _StudentFunction(torch.autograd.Function):
    """One autograd node for the whole student: (x, *trainable params) -> (logits, *hints)."""

    @staticmethod
    def forward(ctx, engine, x, *params):
        ctx.set_materialize_grads(False)
        logits, hints = engine.forward(x)
        ctx.engine, ctx.params = engine, params
        outs = (logits.permute(0, 3, 1, 2),) + tuple(h.permute(0, 3, 1, 2) for h in hints)
        return outs

    @staticmethod
    def backward(ctx, g_logits, *g_hints):
        if g_logits is not None:
            raise EngineError("gradients w.r.t. the student logits are not implemented: the reference back-propagates the "
                              "hint loss only (trainer/layerwise_trainer.py:233-235)")
        grads = ctx.engine.backward(list(g_hints))
        return (None, None) + tuple(grads.get(p) for p in ctx.params)


def run_student(engine, x):
    """Differentiable student call: returns (logits NCHW-logical fp32, [hints NCHW-logical])."""
    params = tuple(p for p in engine.net.parameters() if p.requires_grad)
    if torch.is_grad_enabled() and params:
        outs = _StudentFunction.apply(engine, x, *params)
    else:
        logits, hints = engine.forward(x)
        engine._tape = None
        outs = (logits.permute(0, 3, 1, 2),) + tuple(h.permute(0, 3, 1, 2) for h in hints)
    return outs[0], list(outs[1:])
