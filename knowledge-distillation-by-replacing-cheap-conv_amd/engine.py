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
    writes into one wide buffer (ASPP 5 x 256 ch, decoder 304 -> 320 ch);
  * backward is hand-scheduled over the recorded tape and driven by `requires_grad`: a tensor's
    gradient is computed only when a trainable parameter lies upstream of it, a parameter's only when
    it is trainable.  The reference-faithful plan (only cheap-conv blocks train, loss = hint loss:
    SURVEY F1/F4, "mode A") therefore visits just the sub-graph between the first trainable block
    and the last hint; with `loss = kd + hint` and every student parameter trainable (SURVEY 8d
    "mode B") the same code walks the whole network: logits -> bilinear/decoder/ASPP -> trunk ->
    pools -> stem, with dense-conv / BN(eval) / stem weight gradients.  d(ReLU o BN)/dx is the dgrad
    kernels' mask epilogue; eval-BN weight/bias gradients come from two channel sums of the masked
    gradient against the saved activation;
  * hint features are captured by name at plan time (no Python hooks), reproducing the reference's
    aliasing: a hint on the last conv of a residual block -- or on its `convs` Sequential, or on the
    block itself -- is the block output; `aspp` is the module's concatenated output.
Parameters are read from the nn.Module tree (fp32 masters, reference checkpoint keys); packed
operands are cached per parameter version, so frozen weights are packed once.
"""
import itertools
import os

import torch
from torch import nn

from . import ops
from ._lib import KD_PACK_DGRAD, KD_PACK_FWD
from .lazy import LazyLogits
from .models.students.transform_blocks import DepthwiseSeparableBlock, GateLayer
from .models.wider_resnet import IdentityResidualBlock

_SMALL_CONV = os.environ.get("KDCC_SMALL_CONV", "1") != "0"  # A/B: 0 = GSCNN res2 / res3 zero-padded to 64 channels on the GEMM kernels
_SPLIT_DEC_DGRAD = os.environ.get("KDCC_SPLIT_DEC_DGRAD", "1") != "0"   # A/B: 0 = the decoder's 304-channel input gradient in one launch
_STEM_POOL = os.environ.get("KDCC_STEM_POOL", "1") != "0"   # A/B: 0 = stem conv and pool2 as two kernels
_DW_SUM = os.environ.get("KDCC_DW_SUM", "1") != "0"   # A/B: 0 = one depthwise input-gradient launch per ASPP branch
_CONV_DUAL = os.environ.get("KDCC_CONV_DUAL", "1") != "0"   # A/B: 0 = a bottleneck block's conv3 / proj_conv (and conv1 / proj_conv input gradients) as two launches
_FUSE_BN_SUMS = os.environ.get("KDCC_FUSE_BN_SUMS", "1") != "0"   # A/B: 0 = eval-BN parameter sums by kd_channel_sums only
_FUSE_GAP = os.environ.get("KDCC_FUSE_GAP", "1") != "0"   # A/B: 0 = the ASPP image pooling reads the trunk output again instead of taking the producing conv's channel sums


class EngineError(RuntimeError):
    pass


_FOLD_SERIAL = itertools.count(1)   # identity of a folded (scale, shift) pair: tensor addresses are recycled by the allocator


def _is_trainable(mod):
    return any(p.requires_grad for p in mod.parameters())


def _bn_of(bn_seq):
    return bn_seq[0] if isinstance(bn_seq, nn.Sequential) else bn_seq


def _gate_split(mod):
    """Sequential(block, GateLayer), as TaylorPruneStudent.replace builds it (models/students/taylor_prune_student.py:37-40 of
    the reference) -> (block, gate); anything else -> (mod, None)."""
    if isinstance(mod, nn.Sequential) and len(mod) == 2 and isinstance(mod[1], GateLayer):
        return mod[0], mod[1]
    return mod, None


def _act_gate(bn_seq):
    """The gate behind the ReLU of a bnrelu Sequential(BatchNorm2d, ReLU) -- i.e. Sequential(BN, Sequential(ReLU, GateLayer))."""
    if isinstance(bn_seq, nn.Sequential) and len(bn_seq) > 1:
        return _gate_split(bn_seq[1])[1]
    return None


class _Site:
    """One convolution site of a residual block / ASPP branch: dense nn.Conv2d or a cheap-conv block."""

    def __init__(self, name, mod):
        mod, self.gate = _gate_split(mod)     # a Taylor gate behind the conv: folded into its packed weights
        self.name, self.mod = name, mod
        self.cheap = isinstance(mod, DepthwiseSeparableBlock)
        if self.cheap and self.gate is not None:
            raise EngineError(f"{name}: a gate behind a cheap-conv block is not supported (gates rank the TEACHER's filters)")
        if not self.cheap and not isinstance(mod, nn.Conv2d):
            raise EngineError(f"{name}: unsupported module {type(mod).__name__} in the student graph")
        conv = mod.pointwise_conv if self.cheap else mod
        self.cout = conv.out_channels
        self.cin = (mod.separable_conv if self.cheap else conv).in_channels
        if self.cheap:
            self.k, self.pad, self.dil = mod.geometry
            self.stride = 1
            if mod.separable_conv.bias is not None or mod.pointwise_conv.bias is not None:
                raise EngineError(f"{name}: biased cheap conv inside the fused graph is not supported")
        else:
            if conv.bias is not None or conv.groups != 1:
                raise EngineError(f"{name}: only bias-free dense convs are supported")
            self.k, self.pad, self.dil, self.stride = conv.kernel_size[0], conv.padding[0], conv.dilation[0], conv.stride[0]
        self.trainable = _is_trainable(mod) or (self.gate is not None and self.gate.weight.requires_grad)


class StudentEngine:
    def __init__(self, net, dtype=torch.bfloat16):
        self.net = net
        self.dtype = dtype
        self.hint_names = []
        self._pack = {}   # (id(param), tag) -> (version, packed tensor, param)
        self._bn = {}     # id(bn) -> (version, (scale, shift), bn)
        self._tape = None
        self.last_hint_names = []
        self.reducer = None   # optional parallel.GradReducer: gradients are written into its buckets and announced
        # Taylor importance probes (models/students/taylor_prune_student.py:58-66): for a conv site named here backward also
        # returns d loss / d gate of a unit channel gate placed behind the conv, sum_{n,h,w} y * dL/dy, in probe_grads[name]
        self.probe_names = []
        self.probe_grads = {}
        self._probe_active = False   # set by run_student for differentiable calls (Function.forward runs with grad mode off)
        self._probes = set()
        # Gated-SCNN (models/gscnn.py): the same trunk + a full-resolution shape stream whose edge attention feeds one more
        # ASPP branch; decoder `final_seg`, last upsample without align_corners
        self.is_gscnn = net is not None and hasattr(net, "gate1")
        self._gate_prm = {}
        self._persist = {}   # zero-padded buffers that live across steps (_zero_padded)
        self.edge_prior = None   # (N,H,W) fp32 Canny map handed in by the caller (teacher and student share one per batch)
        self.exported = None     # inputs of block k kept for the student (forward(export_at=k))
        self._next_prefix = None
        # Gated-SCNN: gradients enter the shape stream only through the logits (a loss term on them) or an `aspp` hint -- never
        # under the shipped plan's hint losses.  The owner says which it is (DepthwiseStudent.logits_need_grad, set by trainers
        # that back-propagate a logit loss): the stream then runs on the general kernels and keeps its intermediates.
        self.logits_need_grad = False
        # full-resolution logits materialised on demand (lazy.py); A/B: KDCC_LAZY_LOGITS=0
        self.lazy_logits = os.environ.get("KDCC_LAZY_LOGITS", "1") != "0"
        self._lazy_call = False    # set by the callers that hand the logits to a trainer (run_student, the frozen-teacher call):
                                   # StudentEngine.forward itself keeps returning the materialised (N,H,W,C) tensor
        self._differentiable = False

    def compute_edge_prior(self, x_nchw):
        """The Canny prior of the batch: the device kernel, or the model's `canny_fn` plug point (tests feed the goldens' map)."""
        fn = getattr(self.net, "canny_fn", None)
        N, _, H, W = x_nchw.shape
        if fn is None:
            return ops.canny(x_nchw.detach().float().contiguous())
        return fn(x_nchw).reshape(N, H, W).float().contiguous()

    def _final(self):
        return self.net.final_seg if self.is_gscnn else self.net.final

    def _aspp_lead(self, aspp):
        """Leading non-conv slices of the ASPP concat: image pooling (+ the edge branch of GSCNN)."""
        return 2 if hasattr(aspp, "edge_conv") else 1

    def _grad_like(self, p):
        if self.reducer is not None:
            return self.reducer.grad_buffer(p)
        return torch.empty_like(p, dtype=torch.float32, memory_format=torch.contiguous_format)

    def _grad_done(self, p):
        if self.reducer is not None:
            self.reducer.grad_ready(p)

    # ------------------------------------------------------------------ parameter operands
    def _packed(self, p, tag, fn, extra=None):
        # keyed by id(p), but the entry holds p itself: a Parameter freed by a later replace() cannot hand its id (and a
        # matching small _version) to a new one while the entry is alive, and `ent[2] is p` catches any other aliasing
        key = (id(p), tag)
        ent = self._pack.get(key)
        if ent is not None:
            first = ent[1][0] if isinstance(ent[1], tuple) else ent[1]
        ver = p._version if extra is None else (p._version, extra)   # extra: e.g. the version of a gate folded into the pack
        if ent is None or ent[2] is not p or ent[0] != ver or first.device != p.device:
            ent = (ver, fn(), p)
            self._pack[key] = ent
        return ent[1]

    def drop_caches(self):
        """Forget packed weights / folded BN vectors (called when the module tree is edited: replace(), reset())."""
        self._pack.clear()
        self._bn.clear()
        self._persist.clear()

    def _w_fwd(self, conv, cin_pad=None, cin_rot=0, cout_pad=None, gate=None):
        """cin_rot = r: the conv reads its input channels rotated left by r (engine buffer order [r:], [:r]) -- the decoder keeps
        its concat as [upsampled | fine] so that the 512-B-per-pixel upsample output starts on a 128-B line.  cout_pad: zero
        filters appended up to that many outputs (bot_fine writes the concat buffer's pad channels as zeros itself)."""
        def make():
            w = conv.weight if not cin_rot else torch.cat([conv.weight.detach()[:, cin_rot:], conv.weight.detach()[:, :cin_rot]], 1).contiguous()
            if gate is not None:   # y * g[c] == conv with the filters of output channel c scaled by g[c]
                gv = gate.weight.detach().float()
                if not bool((gv != 0).all()):   # one host sync per gate update (the pack is cached on the gate's version)
                    raise EngineError("a gate folded into a conv's weights has a zero entry: d loss / d gate is recovered from the "
                                      "gated output by a division (_probe), which a zero gate makes undefined (gates start at 1 and "
                                      "the reference's optimizer moves them by ~1e-5 per step)")
                w = w.detach().float() * gv.view(-1, 1, 1, 1)
            if cout_pad is not None and cout_pad > w.shape[0]:
                wp = torch.zeros((cout_pad,) + tuple(w.shape[1:]), dtype=w.dtype, device=w.device)
                wp[:w.shape[0]] = w.detach()
                w = wp
            return ops.pack_conv_weight(w, self.dtype, KD_PACK_FWD, cin_pad)
        return self._packed(conv.weight, ("fwd", self.dtype, cin_pad, cin_rot, cout_pad, id(gate) if gate is not None else None), make,
                            extra=gate.weight._version if gate is not None else None)

    def _w_dgrad(self, conv, cout_pad=None, cin_rot=0, gate=None):
        """[Cin][flipped taps][Cout] operand of the input-gradient conv; cout_pad zero-fills the contraction axis up to the
        GEMM's K granule (the 19-class classifier, the 48-channel bot_fine)."""
        def make():
            w = conv.weight.detach()
            if gate is not None:
                w = w.float() * gate.weight.detach().float().view(-1, 1, 1, 1)
            if cin_rot:
                w = torch.cat([w[:, cin_rot:], w[:, :cin_rot]], 1).contiguous()
            if cout_pad is not None and cout_pad != w.shape[0]:
                wp = torch.zeros((cout_pad,) + tuple(w.shape[1:]), dtype=w.dtype, device=w.device)
                wp[:w.shape[0]] = w
                w = wp
            return ops.pack_conv_weight(w, self.dtype, KD_PACK_DGRAD)
        return self._packed(conv.weight, ("dgrad", self.dtype, cout_pad, cin_rot, id(gate) if gate is not None else None), make,
                            extra=gate.weight._version if gate is not None else None)

    def _w_cat(self, conv_a, conv_b, mode):
        """Packed operand of the K-concatenated 1x1 conv [a | b] (ops.conv2d(..., x2=...)): mode "fwd" -- conv_a and conv_b map
        different inputs onto the SAME output (a bottleneck block's conv3 and proj_conv): [Cout][Cin_a + Cin_b]; mode "dgrad" -- they
        read the SAME input (conv1 and proj_conv): the input gradient's operand [Cin][Cout_a + Cout_b]."""
        def make():
            wa = (self._w_fwd if mode == "fwd" else self._w_dgrad)(conv_a)
            wb = (self._w_fwd if mode == "fwd" else self._w_dgrad)(conv_b)
            return torch.cat([wa, wb], dim=3).contiguous()
        return self._packed(conv_a.weight, ("cat", mode, self.dtype, id(conv_b.weight)), make, extra=conv_b.weight._version)

    def _dual_ok(self, a_shape, cin_a, cin_b, cout, operands, conv_a, conv_b, site=None):
        """Can two 1x1 convs onto / from one tensor run as one K-concatenated launch (bf16, stride 1, plain weights)?"""
        if self.dtype != torch.bfloat16 or not _CONV_DUAL:
            return False
        for c in (conv_a, conv_b):
            if not isinstance(c, nn.Conv2d) or c.kernel_size != (1, 1) or c.stride != (1, 1) or c.padding != (0, 0) or c.bias is not None:
                return False
        if site is not None and (site.cheap or site.gate is not None or site.name in self._probes):
            return False
        N, H, W = a_shape[:3]
        return ops.conv1x1_dual_ok_dims(N, H, W, cin_a, cin_b, cout, self.dtype, operands)

    def _w_dw(self, conv, flip):
        return self._packed(conv.weight, ("dw", flip), lambda: ops.pack_dw_weight(conv.weight, flip))

    def _bn_fold(self, bn_seq):
        """bnrelu Sequential(BatchNorm2d, ReLU) or a bare BatchNorm2d -> cached (scale, shift)."""
        bn = _bn_of(bn_seq)
        if bn.training:
            raise EngineError("the fused student graph implements eval-mode BatchNorm only (LayerwiseTrainer keeps the "
                              "student in eval mode, SURVEY F3)")
        key = id(bn)
        ver = (bn.weight._version, bn.bias._version, bn.running_mean._version, bn.running_var._version, bn.weight.device)
        ent = self._bn.get(key)
        if ent is None or ent[2] is not bn or ent[0] != ver:
            ent = (ver, ops.bn_fold(bn), bn)
            ent[1][0]._kd_serial = next(_FOLD_SERIAL)
            self._bn[key] = ent
        return ent[1]

    def _act_fold(self, bn, gate):
        """(scale, shift) of the epilogue relu(y * scale + shift) for BN -> ReLU -> gate: relu(bn(y)) * g == relu(bn(y) * g)
        for g > 0, so the gate multiplies the folded scale and shift.  gate None: the plain BN fold."""
        scale, shift = self._bn_fold(bn)
        if gate is None:
            return scale, shift
        key = ("gated", id(_bn_of(bn)), id(gate))
        ver = (scale._kd_serial, gate.weight._version, gate.weight.device)
        ent = self._bn.get(key)
        if ent is None or ent[0] != ver or ent[2] is not gate:
            g = gate.weight.detach().float()
            if not bool((g > 0).all()):     # one host sync per gate update; Taylor ranking is not the measured path
                raise EngineError("a gate behind a ReLU must stay positive to be folded into the BN+ReLU epilogue "
                                  "(gates start at 1 and the reference's optimizer moves them by ~1e-5 per step)")
            ent = (ver, ((scale * g).contiguous(), (shift * g).contiguous()), gate)
            ent[1][0]._kd_serial = next(_FOLD_SERIAL)
            self._bn[key] = ent
        return ent[1]

    def _new(self, N, H, W, C, dtype=None, zero=False):
        f = torch.zeros if zero else torch.empty
        return f((N, H, W, C), dtype=dtype or self.dtype, device=self.device)

    # ------------------------------------------------------------------ plan-time checks
    def check_hint_names(self, names):
        """Raise at registration time (epoch 1 / plan time, not 15 epochs into a run) when a hint name is not one the fused
        graph can capture: a conv site `modN.blockM.convs.convK`, a block's `modN.blockM.convs` or `modN.blockM` (both the
        block output), an ASPP branch conv `aspp.features.N.0`, or `aspp` (the module output)."""
        ok = {f"aspp.features.{i}.0" for i in range(len(self.net.aspp.features))}
        ok.add("aspp")   # (GSCNN: the edge branch's gradient re-enters the trunk through the shape stream, _shape_stream_bwd)
        for name, blk in self._flat_blocks():
            ok.add(name)
            ok.add(f"{name}.convs")
            ok.update(f"{name}.convs.{n}" for n, _ in blk.convs.named_children() if n.startswith("conv"))
        bad = [n for n in names if n not in ok]
        if bad:
            raise EngineError(f"hint layers the fused student graph cannot capture: {bad} (supported: conv sites, "
                              "`<block>.convs`, `<block>`, `aspp.features.N.0`, `aspp`)")

    # ------------------------------------------------------------------ forward
    def _flat_blocks(self):
        net = self.net
        flat = [(f"mod{i}.{bn}", blk) for i in range(2, 8) for bn, blk in getattr(net, f"mod{i}").named_children()]
        for name, blk in flat:
            if not isinstance(blk, IdentityResidualBlock):
                raise EngineError(f"{name}: expected an IdentityResidualBlock")
        return flat

    # ------------------------------------------------------------------ frozen-prefix sharing (opt-in)
    def shareable_prefix(self, teacher_engine):
        """Number of leading residual blocks (after the stem / pool2) this student shares with the teacher: same module
        types and bit-identical, frozen parameters and buffers -- which is how DepthwiseStudent builds the student (a deep copy
        of the frozen teacher) up to the first replaced or unfrozen block.  Activations of that prefix are then
        identical in both networks and can be computed once.  Hinted / probed blocks end the prefix."""
        if self.is_gscnn or teacher_engine.is_gscnn or self.dtype != teacher_engine.dtype:
            return 0
        sn, tn = self.net, teacher_engine.net
        key = tuple((id(p), p._version, p.requires_grad) for p in sn.parameters()) + tuple(self.hint_names) + tuple(self.probe_names)
        ent = getattr(self, "_prefix_cache", None)
        if ent is not None and ent[0] == key and ent[1] is tn:
            return ent[2]

        def same(a, b):
            sa, sb = a.state_dict(), b.state_dict()
            if list(sa) != list(sb) or any(p.requires_grad for p in a.parameters()):
                return False
            return all(type(x) is type(y) for x, y in zip(a.modules(), b.modules())) and \
                all(sa[k].shape == sb[k].shape and sa[k].device == sb[k].device and torch.equal(sa[k], sb[k]) for k in sa)
        k = 0
        if same(sn.mod1, tn.mod1):
            sflat, tflat = self._flat_blocks(), teacher_engine._flat_blocks()
            named = set(self.hint_names) | set(self.probe_names)
            for (name, sb), (_, tb) in zip(sflat, tflat):
                if any(h == name or h.startswith(name + ".") for h in named) or not same(sb, tb):
                    break
                k += 1
            # block k itself starts from a tensor the previous epilogue produced with block k's bn1: that must be shared too
            while k > 0 and k < len(sflat) and not same(sflat[k][1].bn1, tflat[k][1].bn1):
                k -= 1
        self._prefix_cache = (key, tn, k)
        return k

    def forward(self, x, collect_hints=True, export_at=None, prefix=None):
        """x: (N,3,H,W) fp32.  Returns (logits (N,H,W,19) fp32 NHWC, [hint tensors (N,h,w,C) NHWC] in forward order).
        export_at = k: also keep the inputs of residual block k in self.exported (the teacher's side of prefix sharing);
        prefix: such an export from the teacher -- the stem, the pools and the blocks before k are not recomputed."""
        net = self.net
        if x.dim() != 4 or x.shape[1] != 3:
            raise EngineError(f"expected an (N,3,H,W) batch, got {tuple(x.shape)}")
        if x.shape[2] % 8 or x.shape[3] % 8:
            raise EngineError("input height/width must be multiples of 8 (two stride-2 pools and one stride-2 conv)")
        self.device = x.device
        N, _, H, W = x.shape
        want = set(self.hint_names) if collect_hints else set()
        self._probes = set(self.probe_names) if self._probe_active else set()
        seen = []
        hints = []
        tape = {"blocks": [], "aspp": None, "hint_slots": [], "pools": {}}

        def note_hint(name, tensor, slot):
            # slot: ("block", block_index, site_index | "out"), ("aspp", branch) or ("aspp_out",)
            seen.append(name)
            hints.append(tensor)
            tape["hint_slots"].append(slot)

        xin = x.detach()
        if xin.dtype != torch.float32 or not xin.is_contiguous():
            xin = xin.float().contiguous()
        flat = self._flat_blocks()
        start = 0
        if prefix is None:
            stem = net.mod1.conv1
            stem_w = stem.weight.detach()
            stem_w = stem_w if stem_w.is_contiguous() else stem_w.contiguous()
            rg = stem.weight.requires_grad
            sc1, sh1 = self._bn_fold(flat[0][1].bn1)   # pool2 is followed by bn1 of mod2.block1
            if _STEM_POOL and self.dtype == torch.bfloat16 and not rg and not self.is_gscnn:
                # frozen stem, no shape stream: pool2 is the only reader of mod1's full-resolution output -- one fused pass
                s = None
                _, a = ops.stem_conv_pool(xin, stem_w, sc1, sh1, want_raw=False)
            else:
                s = ops.stem_conv(xin, stem_w, self.dtype)
                _, a = ops.maxpool3x3s2(s, sc1, sh1, want_raw=False)
            tape["stem"] = dict(x=xin, s=s, rg=rg)
            x_raw = None
            m2 = None
        else:   # the teacher computed everything up to block `start` on identical frozen weights
            start, a, x_raw, m2, s, rg = prefix["k"], prefix["a"], prefix["x_raw"], prefix["m2"], None, False
            tape["blocks"] = [None] * start
            if prefix["pool3"] is not None:
                tape["pools"]["pool3"] = dict(prefix["pool3"], rg=False)
        self.exported = None
        for bi, (name, blk) in enumerate(flat):
            if bi < start:
                continue
            if export_at is not None and bi == export_at:
                self.exported = dict(k=bi, a=a, x_raw=x_raw, m2=m2, pool3=tape["pools"].get("pool3"))
            last_of_mod2 = name.startswith("mod2.") and (bi + 1 == len(flat) or not flat[bi + 1][0].startswith("mod2."))
            is_last = bi + 1 == len(flat)
            nxt = None if (is_last or last_of_mod2) else flat[bi + 1][1]
            need_raw = is_last or last_of_mod2 or (nxt is not None and not hasattr(nxt, "proj_conv"))
            mod_end = is_last or flat[bi + 1][0].split(".")[0] != name.split(".")[0]
            if self.is_gscnn and mod_end and name.split(".")[0] in ("mod3", "mod4"):
                need_raw = True   # dsn3 / dsn4 read the raw module outputs
            # the trunk's last block produces the tensor the ASPP image-pooling branch averages: its conv epilogue takes the per-channel
            # sums where the kernel can (ops.conv2d(out_sums=)), and the pooling never reads the 4096-channel map
            gap = [] if (is_last and _FUSE_GAP and self.dtype == torch.bfloat16) else None
            x_raw, a, rg, rec = self._block_fwd(name, blk, x_raw, a, rg, nxt.bn1 if nxt is not None else None, need_raw,
                                                want, note_hint, bi, out_sums=gap)
            if gap:
                tape["x7_sums"] = gap[0]
            tape["blocks"].append(rec)
            if self.is_gscnn and mod_end:
                tape.setdefault("mods", {})[name.split(".")[0]] = x_raw
                tape.setdefault("mod_end", {})[name.split(".")[0]] = (bi, rg)
            if last_of_mod2:
                m2 = x_raw
                sc, sh = self._bn_fold(flat[bi + 1][1].bn1)
                _, a = ops.maxpool3x3s2(m2, sc, sh, want_raw=False)  # pool3 + bn1 of mod3.block1
                tape["pools"]["pool3"] = dict(x=m2, after=bi, rg=rg)
                x_raw = None
        x7, rg7 = x_raw, rg

        if self.is_gscnn:
            train_shape = self._differentiable and (self.logits_need_grad or "aspp" in want)
            tape["shape"] = {} if train_shape else None
            tape["acts"] = self._shape_stream(xin, s, tape["mods"]["mod3"], tape["mods"]["mod4"], x7, rec=tape["shape"])
        cat, rg_cat = self._aspp_fwd(net.aspp, x7, rg7, want, note_hint, tape)
        logits = self._decoder_fwd(cat, rg_cat, m2, tape["pools"]["pool3"]["rg"], (H, W), tape)

        missing = want - set(seen)
        if missing:
            raise EngineError(f"hint layers not found in the student graph: {sorted(missing)}")
        self._tape = tape
        self.last_hint_names = list(seen)   # forward-execution order (what hooks would have produced)
        return logits, hints

    def _aspp_fwd(self, aspp, x7, rg7, want, note_hint, tape):
        """ASPP (deeplabv3.py:64-75): branches write their channel slices of one concat buffer, image branch first."""
        N, h8, w8, _ = x7.shape
        red = aspp.img_conv[0].out_channels
        nb = len(aspp.features)
        lead = self._aspp_lead(aspp)
        cat = self._new(N, h8, w8, red * (nb + lead))
        sc, sh = self._bn_fold(aspp.img_conv[1])
        ops.aspp_image_pool(x7, aspp.img_conv[0].weight.detach(), sc, sh, cat[..., 0:red], sums=tape.get("x7_sums"))
        if lead == 2:   # edge branch: resampled edge attention -> 1x1 (1 -> red) -> BN -> ReLU
            sc, sh = self._bn_fold(aspp.edge_conv[1])
            ops.edge_aspp(tape["acts"], aspp.edge_conv[0].weight.detach().float().reshape(-1).contiguous(), sc, sh, cat[..., red:2 * red])
        arec = {"x7": x7, "rg7": rg7, "branches": [], "cat": cat, "red": red, "mod": aspp, "lead": lead}
        rg_cat = rg7 or _is_trainable(aspp.img_conv)
        # the replaced branches' depthwise convs all read x7: one pass over it per geometry computes (up to three of) them
        mids, fan = {}, {}
        if _DW_SUM:
            for i, br in enumerate(aspp.features):
                site = _Site(f"aspp.features.{i}.0", br[0])
                if site.cheap:
                    fan.setdefault((site.k, site.pad, site.dil), []).append((i, site))
            for (k, pad, dil), members in fan.items():
                if len(members) > 1:
                    outs = ops.dwconv_fanout(x7, [self._w_dw(st.mod.separable_conv, False) for _, st in members], k, pad, dil)
                    mids.update({i: o for (i, _), o in zip(members, outs)})
        for i, br in enumerate(aspp.features):
            site = _Site(f"aspp.features.{i}.0", br[0])
            act_gate = _gate_split(br[2])[1] if len(br) > 2 else None       # `aspp.features.N.2`: a gate behind the branch's ReLU
            sc, sh = self._act_fold(br[1], act_gate)
            hinted = site.name in want
            probed = site.name in self._probes or site.gate is not None
            raw = self._new(N, h8, w8, red) if (hinted or probed) else None
            out = cat[..., red * (i + lead):red * (i + lead + 1)]
            mid = None
            if site.cheap:
                mid = mids.get(i)
                if mid is None:
                    mid = ops.dwconv(x7, self._w_dw(site.mod.separable_conv, False), site.k, site.pad, site.dil)
                ops.conv2d(mid, self._w_fwd(site.mod.pointwise_conv), out_raw=raw, out_act=out, act_scale=sc, act_shift=sh,
                           act_relu=True)
            else:
                ops.conv2d(x7, self._w_fwd(site.mod, gate=site.gate), 1, site.pad, site.dil, out_raw=raw, out_act=out, act_scale=sc,
                           act_shift=sh, act_relu=True)
            if hinted:
                note_hint(site.name, raw, ("aspp", i))
            arec["branches"].append({"site": site, "mid": mid, "bn": br[1], "probe_raw": raw if probed else None, "act_gate": act_gate})
            rg_cat = rg_cat or site.trainable or _is_trainable(br[1]) or probed or act_gate is not None
        if "aspp" in want:
            note_hint("aspp", cat, ("aspp_out",))
        tape["aspp"] = arec
        return cat, rg_cat

    # ------------------------------------------------------------------ Gated-SCNN shape stream (frozen: forward only)
    def _w_bn_folded(self, conv, bn, cpad):
        """3x3 conv followed by eval-mode BN, as one packed operand: w * scale[co], zero-padded to cpad x cpad channels (the
        shape stream's 32 / 16 channels live in 64-channel buffers so the MFMA kernels can take them), + the padded shift."""
        def make():
            scale, shift = self._bn_fold(bn)
            w = conv.weight.detach().float() * scale.view(-1, 1, 1, 1)
            wp = torch.zeros((cpad, cpad) + tuple(w.shape[2:]), dtype=torch.float32, device=w.device)
            wp[:w.shape[0], :w.shape[1]] = w
            sp = torch.zeros(cpad, dtype=torch.float32, device=w.device)
            sp[:shift.numel()] = shift
            return (ops.pack_conv_weight(wp, self.dtype, KD_PACK_FWD), sp)
        bnm = _bn_of(bn)
        return self._packed(conv.weight, ("bnfold", self.dtype, cpad, id(bn)), make,
                            extra=(bnm.weight._version, bnm.bias._version, bnm.running_mean._version, bnm.running_var._version))

    def _basic_block(self, blk, x, cpad=64, rec=None):
        """Resnet.BasicBlock (encoders/Resnet.py:64-99) on a cpad-channel buffer: two row-buffer 3x3 convs, BN folded into the
        weights, bias + ReLU (+ identity shortcut) in the epilogues.  bf16 blocks of 16 / 32 / 64 channels (res3, res2, res1) run
        on kd_conv3x3_small instead: dense C-channel input and intermediate, the result in the first C channels of a 64-channel
        buffer whose pad channels were zeroed once (its reader is a 1x1 conv padded to the 64-channel GEMM granule).
        rec (a list): the stream is being differentiated -- general kernels only, nothing in persistent buffers, and
        (blk, x, t, y) is appended for _basic_block_bwd."""
        if rec is None:
            for p in blk.parameters():
                if p.requires_grad:
                    raise EngineError("trainable shape-stream parameters need the differentiable shape stream: set "
                                      "DepthwiseStudent.logits_need_grad = True (trainers that back-propagate a logit loss do)")
        N, H, W, cx = x.shape
        planes = blk.conv1.out_channels
        if rec is None and _SMALL_CONV and self.dtype == torch.bfloat16 and planes in (16, 32, 64) and blk.conv1.in_channels == planes:
            w1, s1 = self._w_bn_folded(blk.conv1, blk.bn1, planes)
            w2, s2 = self._w_bn_folded(blk.conv2, blk.bn2, planes)
            xin = x if cx == planes else x[..., :planes]
            t = self._new(N, H, W, planes)
            ops.conv3x3_small(xin, w1, s1, relu=True, out=t)
            y = self._zero_padded(("basic_block", id(blk)), N, H, W, cpad) if planes < cpad else self._new(N, H, W, planes)
            ops.conv3x3_small(t, w2, s2, res=xin, relu=True, out=y[..., :planes])
            return y
        w1, s1 = self._w_bn_folded(blk.conv1, blk.bn1, cpad)
        w2, s2 = self._w_bn_folded(blk.conv2, blk.bn2, cpad)
        t = self._new(N, H, W, cpad)
        ops.conv2d(x, w1, 1, 1, 1, out_act=t, act_shift=s1, act_relu=True)
        y = self._new(N, H, W, cpad)
        ops.conv2d(t, w2, 1, 1, 1, res_pre=x, out_act=y, act_shift=s2, act_relu=True)
        if rec is not None:
            rec.append((blk, x, t, y))
        return y

    def _zero_padded(self, tag, N, H, W, C):
        """A buffer that persists across steps and was zero-filled ONCE: callers write the same leading channels every step and
        rely on the rest staying zero (a per-step fill of these full-resolution buffers was 12 x 0.14 ms of a GSCNN step).
        Only for tensors consumed inside the forward that produced them (the frozen shape stream keeps nothing for backward)."""
        key = (tag, N, H, W, C, self.dtype)
        buf = self._persist.get(key)
        if buf is None or buf.device != self.device:
            buf = torch.zeros((N, H, W, C), dtype=self.dtype, device=self.device)
            self._persist[key] = buf
        return buf

    def _gate_params(self, gate):
        """GatedSpatialConv2d -> the packed fp32 vector kd_gated_conv takes (both eval-mode BNs folded)."""
        ent = self._gate_prm.get(id(gate))
        ver = tuple(t._version for t in list(gate.parameters()) + list(gate.buffers()))     # trainable gates change every step
        if ent is not None and ent[0] is gate and ent[2] == ver:
            return ent[1]
        bn0, c1, _, c2, bn4, _ = gate._gate_conv
        hdim = c1.in_channels
        s0, t0 = self._bn_fold(bn0)
        s4, t4 = self._bn_fold(bn4)
        w1 = c1.weight.detach().float().view(hdim, hdim)
        prm = torch.cat([(w1 * s0.view(1, -1)).reshape(-1), c1.bias.detach().float() + w1 @ t0,
                         c2.weight.detach().float().view(hdim) * s4, c2.bias.detach().float() * s4 + t4,
                         gate.weight.detach().float().reshape(-1)]).contiguous()
        self._gate_prm[id(gate)] = (gate, prm, ver)
        return prm

    def _w_bn_folded_dgrad(self, conv, bn, cpad):
        """Input-gradient operand of a BN-folded 3x3 conv (see _w_bn_folded): [Cin][flipped taps][Cout] of w * scale[co], padded."""
        def make():
            scale, _ = self._bn_fold(bn)
            w = conv.weight.detach().float() * scale.view(-1, 1, 1, 1)
            wp = torch.zeros((cpad, cpad) + tuple(w.shape[2:]), dtype=torch.float32, device=w.device)
            wp[:w.shape[0], :w.shape[1]] = w
            return ops.pack_conv_weight(wp, self.dtype, KD_PACK_DGRAD)
        bnm = _bn_of(bn)
        return self._packed(conv.weight, ("bnfold_dgrad", self.dtype, cpad, id(bn)), make,
                            extra=(bnm.weight._version, bnm.bias._version, bnm.running_var._version))

    def _shape_stream(self, x_nchw, m1, m3, m4, m7, rec=None):
        """gscnn.py:269-314: edge attention `acts` (N,H,W) fp32 from the stem output, the three side outputs and the Canny
        prior.  The reference's interpolations of full-resolution tensors to the full resolution are identities.
        rec (a dict): the stream is being differentiated (see forward); every layer's operands are recorded for
        _shape_stream_bwd and the bf16-only small-channel kernels / persistent buffers are not used."""
        net = self.net
        N, H, W, _ = m1.shape
        train = rec is not None
        if train:
            rec.update(sides=[], blocks=[], squeezes=[], gates=[], m1=m1)

        def side(conv, t):   # dsnK: 1x1 (C -> 1) + bias, then bilinear (align_corners) to the input size
            o = self._new(N, t.shape[1], t.shape[2], 1)
            ops.conv2d(t, self._w_fwd(conv), out_act=o, act_shift=conv.bias.detach().float().contiguous())
            smap = ops.upsample_bilinear_ac(o, (H, W))
            if train:
                rec["sides"].append((conv, t))
            return smap

        small = _SMALL_CONV and self.dtype == torch.bfloat16 and not train

        def squeeze(conv, x, cout):   # dK: 1x1 + bias -> `cout` dense channels (its only reader is the gated conv)
            if small and (conv.in_channels, cout) in ((64, 32), (32, 16), (16, 8)) and x.shape[3] == conv.in_channels:
                return ops.pointwise_small(x, conv.weight.detach().float().reshape(cout, -1).contiguous(),
                                           conv.bias.detach().float().contiguous())
            buf = self._new(N, H, W, cout)
            ops.conv2d(x, self._w_fwd(conv, cin_pad=64 if conv.in_channels < 64 else None), out_act=buf,
                       act_shift=conv.bias.detach().float().contiguous())
            if train:
                rec["squeezes"].append((conv, x, buf))
            return buf

        def gated(gate, buf, side_map, c):
            # its reader is a BasicBlock: dense c channels for kd_conv3x3_small, else the first c channels of a 64-channel buffer
            # with zero pad channels (the GEMM kernels read all 64)
            if small:
                return ops.gated_conv(buf, side_map, self._gate_params(gate), c, out=self._new(N, H, W, c))
            out = self._new(N, H, W, 64, zero=True) if train else self._zero_padded(("gated", id(gate)), N, H, W, 64)
            ops.gated_conv(buf, side_map, self._gate_params(gate), c, out=out[..., :c])
            if train:
                rec["gates"].append((gate, buf, side_map, c))
            return out

        blocks = rec["blocks"] if train else None
        s3, s4, s7 = side(net.dsn3, m3), side(net.dsn4, m4), side(net.dsn7, m7)
        # (small: the BasicBlocks hand dense C-channel tensors to kd_pointwise_small; else 64-channel buffers to the GEMM kernels)
        cs = gated(net.gate1, squeeze(net.d1, self._basic_block(net.res1, m1, rec=blocks), 32), s3, 32)
        cs = gated(net.gate2, squeeze(net.d2, self._basic_block(net.res2, cs, 32 if small else 64, rec=blocks), 16), s4, 16)
        cs = gated(net.gate3, squeeze(net.d3, self._basic_block(net.res3, cs, 16 if small else 64, rec=blocks), 8), s7, 8)
        canny = self.edge_prior if self.edge_prior is not None else self.compute_edge_prior(x_nchw)
        w = torch.cat([net.fuse.weight.detach().float().reshape(-1), net.cw.weight.detach().float().reshape(-1)]).contiguous()
        if train:
            rec["edge"] = (cs, canny, w)
        return ops.edge_attention(cs, canny, w)

    # ---- Gated-SCNN shape stream, backward (gscnn.py:269-314 under autograd) ---------------------------------------------------
    def _give(self, p, g, grads):
        """Hand the gradient `g` of parameter p (any shape with p's element count) to autograd / the reducer."""
        if p is None or not p.requires_grad:
            return
        buf = self._grad_like(p)
        buf.copy_(g.reshape(buf.shape))
        grads[p] = buf
        self._grad_done(p)

    def _bn_raw(self, bn):
        """(gamma, beta, mean, sigma) of an eval-mode BatchNorm as fp32 vectors."""
        return (bn.weight.detach().float(), bn.bias.detach().float(), bn.running_mean.float(),
                torch.sqrt(bn.running_var.float() + bn.eps))

    def _basic_block_bwd(self, blk, x, t, y, g_y, grads, need_in):
        """Backward of _basic_block's general path on 64-channel padded buffers.  y = relu(z2 + x), z2 = bn2(conv2(t)),
        t = relu(bn1(conv1(x))), both BNs folded into the convs in the forward.  g_y: (N,H,W,64) gradient (pad channels zero)."""
        C = blk.conv1.out_channels
        ones = self._ones(y.shape[3])
        g_z2 = ops.relu_bn_bwd(g_y, y, ones)                                                  # through the last ReLU
        g_z1 = self._new(*t.shape)
        ops.conv2d(g_z2, self._w_bn_folded_dgrad(blk.conv2, blk.bn2, y.shape[3]), 1, 1, 1, out_raw=g_z1, mask=t)   # dgrad2, then ReLU of t
        for conv, bn, a_in, g_z, act, sub in ((blk.conv2, blk.bn2, t, g_z2, y, x), (blk.conv1, blk.bn1, x, g_z1, t, None)):
            scale, _ = self._bn_fold(bn)
            if conv.weight.requires_grad:
                gw = torch.empty((a_in.shape[3], a_in.shape[3], 3, 3), dtype=torch.float32, device=a_in.device)
                ops.conv2d_wgrad(a_in, g_z, gw, 1, 1, 1)
                self._give(conv.weight, gw[:C, :C] * scale.view(-1, 1, 1, 1), grads)           # the conv ran with w * scale[co]
            if bn.weight.requires_grad or bn.bias.requires_grad:
                # z = bn(c): dbeta = sum g_z, dgamma = sum g_z (z - beta) / gamma, with z = act (- shortcut) where the ReLU passed
                s1, s2 = ops.channel_sums(g_z, a=act)
                if sub is not None:
                    _, s2b = ops.channel_sums(g_z, a=sub)
                    s2 = s2 - s2b
                gamma, beta, _, _ = self._bn_raw(bn)
                self._give(bn.bias, s1[:C], grads)
                self._give(bn.weight, (s2[:C] - beta * s1[:C]) / gamma, grads)
        if not need_in:
            return None
        g_x = self._new(*x.shape)
        ops.conv2d(g_z1, self._w_bn_folded_dgrad(blk.conv1, blk.bn1, x.shape[3]), 1, 1, 1, out_raw=g_x, res_post=g_z2)
        return g_x

    def _ones(self, n):
        key = ("ones", n)
        if key not in self._persist or self._persist[key].device != self.device:
            self._persist[key] = torch.ones(n, dtype=torch.float32, device=self.device)
        return self._persist[key]

    def _gated_conv_bwd(self, gate, feat, side_map, C, g_out, grads):
        """Backward of GatedSpatialConv2d (gate_spatial_conv.py:50-60): returns (g_feat (N,H,W,C) fp32, g_gate (N,H,W) fp32).
        The hidden activations are recomputed from feat / the side map with the folded parameters the forward used."""
        bn0, c1, _, c2, bn4, _ = gate._gate_conv
        Hd = C + 1
        s0, t0 = self._bn_fold(bn0)
        s4, t4 = self._bn_fold(bn4)
        W1 = c1.weight.detach().float().view(Hd, Hd)
        b1 = c1.bias.detach().float()
        w2 = c2.weight.detach().float().view(1, Hd)
        b2 = c2.bias.detach().float()
        W1f, b1f = W1 * s0.view(1, -1), b1 + W1 @ t0
        w2f, b2f = w2 * s4, b2 * s4 + t4
        Wg = gate.weight.detach().float().view(C, C)
        f = feat[..., :C]
        U = torch.cat([f.float(), side_map.float().reshape(f.shape[:3] + (1,))], dim=3)     # [feat; gate] (a copy, no arithmetic)
        Z = ops.small_linear(U, W1f, b1f, relu=True)
        A = ops.small_linear(Z, w2f, b2f).reshape(f.shape[:3])
        g_v = ops.small_linear(g_out[..., :C], Wg.t().contiguous(), out_dtype=torch.float32)
        g_feat, g_a, v = ops.gate_mix_bwd(f, A, gv=g_v, want_v=gate.weight.requires_grad)
        if gate.weight.requires_grad:
            dWg, _ = ops.small_wgrad(v, g_out[..., :C])
            self._give(gate.weight, dWg, grads)
        g_a1 = g_a.unsqueeze(3)
        train_gate = any(p.requires_grad for p in gate._gate_conv.parameters())
        g_H = ops.small_linear(g_a1, w2f.t().contiguous(), mask=Z)                          # through c2, then the ReLU of Z
        if train_gate:
            dw2f, db2f = ops.small_wgrad(Z, g_a1, want_bias=True)                             # (1, Hd), (1,)
            dW1f, db1f = ops.small_wgrad(U, g_H, want_bias=True)                              # (Hd, Hd), (Hd,)
            # folded -> raw parameters: W1f = W1 diag(s0), b1f = b1 + W1 t0; w2f = w2 s4, b2f = b2 s4 + t4;  s = gamma / sigma,
            # t = beta - mean s
            self._give(c1.weight, dW1f * s0.view(1, -1) + db1f.view(-1, 1) * t0.view(1, -1), grads)
            self._give(c1.bias, db1f, grads)
            ds0, dt0 = (dW1f * W1).sum(0), (db1f.view(-1, 1) * W1).sum(0)
            _, _, mu0, sig0 = self._bn_raw(bn0)
            self._give(bn0.weight, (ds0 - mu0 * dt0) / sig0, grads)
            self._give(bn0.bias, dt0, grads)
            self._give(c2.weight, dw2f * s4, grads)
            self._give(c2.bias, db2f * s4, grads)
            ds4, dt4 = (dw2f * w2).sum().view(1) + db2f * b2, db2f
            _, _, mu4, sig4 = self._bn_raw(bn4)
            self._give(bn4.weight, (ds4 - mu4 * dt4) / sig4, grads)
            self._give(bn4.bias, dt4, grads)
        W1ft = W1f.t().contiguous()                                                          # (Hd in, Hd out) -> rows = inputs
        ops.small_linear(g_H, W1ft[:C], out=g_feat, accumulate=True)                         # g_feat += (W1f^T g_H)[:C]
        g_gate = ops.small_linear(g_H, W1ft[C:C + 1]).reshape(f.shape[:3])
        return g_feat, g_gate

    def _shape_stream_bwd(self, g_acts, grads, need):
        """g_acts: (N,H,W) fp32 gradient of the edge attention.  need: {'m1','m3','m4','m7'} -> bool, which trunk tensors have
        trainable parameters upstream.  Returns {name: gradient tensor in the engine dtype | None}."""
        rec = self._tape["shape"]
        net = self.net
        cs, canny, w = rec["edge"]
        N, H, W = g_acts.shape
        g_t, g_s, eo_canny = ops.edge_attention_bwd(cs, canny, w, g_acts.contiguous())
        if net.fuse.weight.requires_grad:
            self._give(net.fuse.weight, ops.small_wgrad(cs[..., :8], g_s.unsqueeze(3))[0], grads)
        if net.cw.weight.requires_grad:
            self._give(net.cw.weight, ops.small_wgrad(eo_canny, g_t.unsqueeze(3))[0], grads)
        g = ops.small_linear(g_s.unsqueeze(3), net.fuse.weight.detach().float().view(1, 8).t().contiguous())   # (N,H,W,8) fp32
        out = {"m1": None, "m3": None, "m4": None, "m7": None}
        side_names = ["m3", "m4", "m7"]
        for k in (2, 1, 0):                                      # gate3/d3/res3 ... gate1/d1/res1
            gate, feat, side_map, C = rec["gates"][k]
            g_feat, g_gate = self._gated_conv_bwd(gate, feat, side_map, C, g, grads)
            # side output: s = upsample(conv1x1(m) + b)
            conv, m = rec["sides"][k]
            nm = side_names[k]
            if need[nm] or conv.weight.requires_grad or conv.bias.requires_grad:
                g_o = ops.upsample_bilinear_ac_bwd(g_gate.unsqueeze(3), (m.shape[1], m.shape[2]))              # (N,h,w,1) fp32
                if conv.weight.requires_grad or conv.bias.requires_grad:
                    pad8 = torch.zeros(g_o.shape[:3] + (8,), dtype=m.dtype, device=m.device)
                    pad8[..., 0:1] = g_o.to(m.dtype)
                    dw8 = torch.empty((8, m.shape[3], 1, 1), dtype=torch.float32, device=m.device)
                    ops.pw_wgrad(m, pad8, dw8)
                    self._give(conv.weight, dw8[0], grads)
                    self._give(conv.bias, ops.channel_sums(pad8)[0][0:1], grads)
                if need[nm]:
                    gm = self._new(*m.shape)
                    ops.rank1_add(gm, g_o.reshape(-1), conv.weight.detach().float().reshape(-1), accumulate=False)
                    out[nm] = gm
            # squeeze dK: 1x1 + bias on the BasicBlock's (64-channel padded) output
            sq, x_sq, _ = rec["squeezes"][k]
            cin = sq.in_channels
            if sq.weight.requires_grad or sq.bias.requires_grad:
                dW, db = ops.small_wgrad(x_sq[..., :cin], g_feat, want_bias=True)
                self._give(sq.weight, dW, grads)
                self._give(sq.bias, db, grads)
            g_y = self._new(N, H, W, x_sq.shape[3], zero=x_sq.shape[3] != cin)
            ops.small_linear(g_feat, sq.weight.detach().float().view(sq.out_channels, cin).t().contiguous(), out=g_y[..., :cin])
            blk, x, t, y = rec["blocks"][k]
            first = k == 0
            g_x = self._basic_block_bwd(blk, x, t, y, g_y, grads, need_in=(not first) or need["m1"])
            if first:
                out["m1"] = g_x
            else:
                g = g_x                                           # gradient of the previous gated conv's (padded) output
        return out

    def _decoder_fwd(self, cat, rg_cat, m2, rg_m2, size, tape):
        """bot_aspp / bot_fine / upsample x4 / final / upsample to the input size (deeplabv3.py:141-162)."""
        net = self.net
        N, h8, w8, _ = cat.shape
        up_small = self._new(N, h8, w8, net.bot_aspp.out_channels)
        ops.conv2d(cat, self._w_fwd(net.bot_aspp), out_raw=up_small)
        h2, w2 = m2.shape[1], m2.shape[2]
        nf, nu = net.bot_fine.out_channels, net.bot_aspp.out_channels
        cdec = nf + nu
        cpad = ((cdec + 63) // 64) * 64          # 48 + 256 = 304 -> 320: the GEMM K granule
        dec0 = self._new(N, h2, w2, cpad)
        # buffer order [upsampled (nu) | fine (nf) | pad]: the reference concatenates [fine, upsampled] (deeplabv3.py:152), the
        # final conv reads its input channels rotated by nf instead.  The upsample writes 512 B per pixel; starting 96 B into
        # a line it took 0.43 ms, line-aligned 0.25 (tools/ubench/upsample_align.py)
        # (bot_fine carries cpad - cdec zero filters: the pad channels, which the final conv multiplies by zero weights, must
        # be finite, and a strided fill of them was a 58-us kernel of its own)
        ops.conv2d(m2, self._w_fwd(net.bot_fine, cout_pad=cpad - nu), out_raw=dec0[..., nu:cpad], algo_cout=nf)
        ops.upsample_bilinear_ac(up_small, (h2, w2), out=dec0[..., 0:nu])
        f = self._final()
        sc, sh = self._bn_fold(f[1])
        d1 = self._new(N, h2, w2, f[0].out_channels)
        ops.conv2d(dec0, self._w_fwd(f[0], cin_pad=cpad, cin_rot=nf), 1, 1, 1, out_act=d1, act_scale=sc, act_shift=sh, act_relu=True,
                   algo_cin=cdec)
        sc, sh = self._bn_fold(f[4])
        ncls = f[6].out_channels
        d3 = self._new(N, h2, w2, ncls, dtype=torch.float32)
        # Where nothing differentiates through the head's last activation (the frozen teacher; the student when the loss is the hints alone),
        # the classifier runs in the epilogue of final[3] and the 256-channel tensor is neither written nor read back (kd_conv_epilogue.cls_w)
        fuse_cls = (not self.logits_need_grad and not any(q.requires_grad for q in list(f[3].parameters()) + list(f[6].parameters())) and
                    f[6].bias is None and ncls <= 32 and f[6].kernel_size == (1, 1) and ops.conv_cls_ok(d1, f[3].out_channels, 3, 1))
        d2 = None
        if fuse_cls:
            try:
                ops.conv2d(d1, self._w_fwd(f[3]), 1, 1, 1, act_scale=sc, act_shift=sh, act_relu=True, cls_w=self._w_fwd(f[6], cout_pad=32), cls_out=d3)
            except ops.ClsUnsupported:
                fuse_cls = False
        if not fuse_cls:
            d2 = self._new(N, h2, w2, f[3].out_channels)
            ops.conv2d(d1, self._w_fwd(f[3]), 1, 1, 1, out_act=d2, act_scale=sc, act_shift=sh, act_relu=True)
            ops.conv2d(d2, self._w_fwd(f[6]), out_raw=d3)
        if self.lazy_logits and self._lazy_call and not self.logits_need_grad:
            # nothing differentiates through the logits: hand out the half-resolution tensor behind the full-resolution
            # signature (lazy.LazyLogits); the logged criteria interpolate in registers, anything else materialises it
            logits = LazyLogits(d3, size, align_corners=not self.is_gscnn)
        else:
            logits = ops.upsample_bilinear_ac(d3, size, out_dtype=torch.float32, align_corners=not self.is_gscnn)
        tape["dec"] = dict(cat=cat, rg_cat=rg_cat, m2=m2, rg_m2=rg_m2, dec0=dec0, d1=d1, d2=d2, cdec=cdec, nf=nf, nu=nu, size=size,
                           small=(h8, w8))
        return logits

    def _block_fwd(self, name, blk, x_raw, a1, rg_in, next_bn, need_raw, want, note_hint, bi, out_sums=None):
        """One pre-activation residual block.  a1 = relu(bn1(x)) already produced by the previous kernel.
        rg_in: a trainable parameter lies upstream of the block input."""
        convs = [(n, m) for n, m in blk.convs.named_children() if n.startswith("conv")]
        bns = {n: m for n, m in blk.convs.named_children() if n.startswith("bn")}
        for n, m in blk.convs.named_children():
            if isinstance(m, (nn.Dropout, nn.Dropout2d)) and m.training:
                raise EngineError(f"{name}: training-mode dropout is not supported (student runs in eval mode, SURVEY F3)")
        sites = [_Site(f"{name}.convs.{n}", m) for n, m in convs]
        N = a1.shape[0]
        rec = {"name": name, "blk": blk, "sites": sites, "a_in": [], "mid": [], "rg_a": [], "hint_raw": [None] * len(sites),
               "x_raw": x_raw, "rg_in": rg_in, "proj": hasattr(blk, "proj_conv")}
        rg_a1 = rg_in or _is_trainable(blk.bn1)      # a1 = relu(bn1(x))
        # bottleneck blocks (mod6 / mod7): conv3 and proj_conv are two 1x1 convs onto the block output -- one K-concatenated launch
        # [conv2's activation | a1] . [W3 | Wp]^T, the shortcut tensor never exists (ops.conv2d(..., x2=))
        dual = (rec["proj"] and len(sites) == 3 and sites[-1].k == 1 and sites[-1].stride == 1 and
                self._dual_ok(a1.shape, sites[-1].cin, a1.shape[3], sites[-1].cout, 0, sites[-1].mod, blk.proj_conv, sites[-1]))
        rec["dual"] = dual
        # shortcut
        if dual:
            shortcut = None
            rg_short = rg_a1 or blk.proj_conv.weight.requires_grad
        elif rec["proj"]:
            pc = blk.proj_conv
            ho = ops.conv_out_size(a1.shape[1], 1, pc.stride[0], 0, 1)
            wo = ops.conv_out_size(a1.shape[2], 1, pc.stride[0], 0, 1)
            shortcut = self._new(N, ho, wo, pc.out_channels)
            ops.conv2d(a1, self._w_fwd(pc), pc.stride[0], 0, 1, out_raw=shortcut)
            rg_short = rg_a1 or pc.weight.requires_grad
        else:
            if x_raw is None:
                raise EngineError(f"{name}: identity shortcut needs the raw block input")
            shortcut = x_raw
            rg_short = rg_in
        a, rg = a1, rg_a1
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
            probed = site.name in self._probes or site.gate is not None
            kw = {}
            if last:
                want_raw = need_raw or hinted or probed
                raw = self._new(N, ho, wo, site.cout) if want_raw else None
                if not dual:
                    kw["res_pre"] = shortcut
                if next_bn is not None:
                    sc, sh = self._bn_fold(next_bn)
                    a_next = self._new(N, ho, wo, site.cout)
                    kw.update(out_act=a_next, act_scale=sc, act_shift=sh, act_relu=True)
                kw["out_raw"] = raw
                x_out = raw
                if out_sums is not None and not site.cheap and site.gate is None:
                    kw["out_sums"] = out_sums
            else:
                sc, sh = self._act_fold(bns[f"bn{i + 2}"], _act_gate(bns[f"bn{i + 2}"]))
                act = self._new(N, ho, wo, site.cout)
                raw = self._new(N, ho, wo, site.cout) if (hinted or probed) else None
                kw.update(out_raw=raw, out_act=act, act_scale=sc, act_shift=sh, act_relu=True)
            if site.cheap:
                mid = ops.dwconv(a, self._w_dw(site.mod.separable_conv, False), site.k, site.pad, site.dil)
                ops.conv2d(mid, self._w_fwd(site.mod.pointwise_conv), **kw)
            elif last and dual:
                mid = None
                try:
                    ops.conv2d(a, self._w_cat(site.mod, blk.proj_conv, "fwd"), x2=a1, **kw)
                except ops.DualUnsupported:
                    # _dual_ok asked for dense operands of these sizes; the real epilogue / views can still be refused (raised
                    # before anything is launched).  kdcc.h's contract: fall back to the two launches.
                    rec["dual"] = False
                    shortcut = self._new(N, ho, wo, site.cout)
                    ops.conv2d(a1, self._w_fwd(blk.proj_conv), 1, 0, 1, out_raw=shortcut)
                    ops.conv2d(a, self._w_fwd(site.mod, gate=site.gate), site.stride, site.pad, site.dil, res_pre=shortcut, **kw)
            else:
                mid = None
                ops.conv2d(a, self._w_fwd(site.mod, gate=site.gate), site.stride, site.pad, site.dil, **kw)
            rec["mid"].append(mid)
            rg = rg or site.trainable or probed
            if probed:
                rec.setdefault("probe", {})[i] = (raw, shortcut if last else None)
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
                rg = rg or _is_trainable(bns[f"bn{i + 2}"])
        rec["rg_out"] = rg or rg_short
        return x_out, a_next, rec["rg_out"], rec

    # ------------------------------------------------------------------ backward
    def grad_production_order(self):
        """Trainable parameters in the order backward produces their gradients: decoder (classifier first), ASPP, then the
        trunk blocks from mod7 back to mod2 (within a block last conv first; pointwise before depthwise; a conv before the
        BN that feeds it), the stem last.  Used to lay out the all-reduce buckets."""
        net, order = self.net, []

        def add_p(*ps):
            for p in ps:
                if p is not None and p.requires_grad and all(p is not q for q in order):
                    order.append(p)

        def add_conv(mod):
            mod, gate = _gate_split(mod)
            if gate is not None:
                add_p(gate.weight)
            if isinstance(mod, DepthwiseSeparableBlock):
                add_p(mod.pointwise_conv.weight, mod.separable_conv.weight)
            else:
                add_p(mod.weight)

        def add_bn(seq, gate=None):
            bn = _bn_of(seq)
            gate = gate if gate is not None else _act_gate(seq)
            if gate is not None:
                add_p(gate.weight)
            add_p(bn.weight, bn.bias)
        f = self._final()
        add_conv(f[6]); add_bn(f[4]); add_conv(f[3]); add_bn(f[1]); add_conv(f[0])
        add_conv(net.bot_fine); add_conv(net.bot_aspp)
        add_bn(net.aspp.img_conv[1]); add_conv(net.aspp.img_conv[0])
        if self.is_gscnn:   # edge branch, then the shape stream from its end (fuse / cw) back to res1, the side outputs with their gate
            add_bn(net.aspp.edge_conv[1]); add_conv(net.aspp.edge_conv[0])
            add_p(net.fuse.weight, net.cw.weight)
            for gate, dsn, d, res in ((net.gate3, net.dsn7, net.d3, net.res3), (net.gate2, net.dsn4, net.d2, net.res2),
                                      (net.gate1, net.dsn3, net.d1, net.res1)):
                add_p(*gate.parameters()); add_p(*dsn.parameters()); add_p(*d.parameters()); add_p(*res.parameters())
        for br in net.aspp.features:
            add_bn(br[1], _gate_split(br[2])[1] if len(br) > 2 else None); add_conv(br[0])
        for _, blk in reversed(self._flat_blocks()):
            items = list(blk.convs.named_children())
            for n, m in reversed(items):
                if n.startswith("conv"):
                    add_conv(m)
                elif n.startswith("bn"):
                    add_bn(m)
            if hasattr(blk, "proj_conv"):
                add_conv(blk.proj_conv)
            add_bn(blk.bn1)
        add_conv(net.mod1.conv1)
        return order

    def backward(self, hint_grads, g_logits=None):
        """hint_grads: list aligned with forward()'s hints; entries are (N,h,w,C) NHWC tensors or None.  g_logits: gradient
        w.r.t. the (N,H,W,classes) logits or None.  Returns {parameter: fp32 gradient} for every trainable parameter reached."""
        tape = self._tape
        if tape is None:
            raise EngineError("backward() called without a recorded forward()")
        if len(hint_grads) != len(tape["hint_slots"]):
            raise EngineError("hint gradient list does not match the recorded hints")
        grads = {}
        self.probe_grads = {}
        g_block_out = {}   # block index -> grad wrt raw block output
        g_site_hint = {}   # (block index, site index) -> hint grad of that site's raw output
        g_aspp = {}        # branch -> grad wrt the branch conv's raw output
        g_cat_hint = None  # grad wrt the ASPP module output (behind BN+ReLU)

        def acc(d, k, g):
            d[k] = g if k not in d else d[k] + g
        for g, slot in zip(hint_grads, tape["hint_slots"]):
            if g is None:
                continue
            g = self._as_nhwc(g)
            if slot[0] == "aspp":
                acc(g_aspp, slot[1], g)
            elif slot[0] == "aspp_out":
                g_cat_hint = g if g_cat_hint is None else g_cat_hint + g
            elif slot[2] == "out":
                acc(g_block_out, slot[1], g)
            else:
                acc(g_site_hint, (slot[1], slot[2]), g)

        nb = len(tape["blocks"])
        g_cat = None
        if g_logits is not None:
            g_cat, g_m2 = self._decoder_bwd(self._as_nhwc(g_logits, torch.float32), grads)
            if g_m2 is not None:
                acc(g_block_out, tape["pools"]["pool3"]["after"], g_m2)
        g_x7 = self._aspp_bwd(g_cat, g_cat_hint, g_aspp, grads)
        if g_x7 is not None:
            g_block_out[nb - 1] = g_x7 if (nb - 1) not in g_block_out else g_block_out[nb - 1].add_(g_x7)
        g_m1 = None
        sg = getattr(self, "_shape_grads", None)
        if sg is not None:     # Gated-SCNN: what the shape stream sends back into the trunk (side outputs dsn3/4/7, the stem output)
            self._shape_grads = None
            for nm, bi_ in (("m7", nb - 1), ("m4", tape["mod_end"]["mod4"][0]), ("m3", tape["mod_end"]["mod3"][0])):
                if sg[nm] is not None:
                    g_block_out[bi_] = sg[nm] if bi_ not in g_block_out else g_block_out[bi_].add_(sg[nm])
            g_m1 = sg["m1"]

        pool3 = tape["pools"]["pool3"]
        for bi in range(nb - 1, -1, -1):
            rec = tape["blocks"][bi]
            if rec is None:     # a block of the shared frozen prefix: nothing trainable lies there
                continue
            g_out = g_block_out.pop(bi, None)
            has_inner = any((bi, i) in g_site_hint for i in range(len(rec["sites"])))
            if g_out is None and not has_inner:
                continue
            g_xin = self._block_bwd(bi, rec, g_out, g_site_hint, grads)
            if g_xin is None:
                continue
            if bi == pool3["after"] + 1:
                g_prev = ops.maxpool3x3s2_bwd(pool3["x"], g_xin)     # pool3: back to mod2's output
                g_block_out[bi - 1] = g_prev if (bi - 1) not in g_block_out else g_block_out[bi - 1].add_(g_prev)
            elif bi == 0:
                st = tape["stem"]                                     # pool2, then the stem conv's weight gradient
                g_s = ops.maxpool3x3s2_bwd(st["s"], g_xin)
                if g_m1 is not None:                                  # + the shape stream's share (res1 reads the stem output)
                    g_s.add_(g_m1[..., :g_s.shape[3]])
                    g_m1 = None
                w = self.net.mod1.conv1.weight
                gw = self._grad_like(w)
                ops.stem_wgrad(st["x"], g_s, gw)
                grads[w] = gw
                self._grad_done(w)
            else:
                g_block_out[bi - 1] = g_xin if (bi - 1) not in g_block_out else g_block_out[bi - 1].add_(g_xin)
        if g_m1 is not None and self.net.mod1.conv1.weight.requires_grad:   # only the shape stream reached the stem
            w = self.net.mod1.conv1.weight
            gw = self._grad_like(w)
            ops.stem_wgrad(tape["stem"]["x"], g_m1[..., :64].contiguous(), gw)
            grads[w] = gw
            self._grad_done(w)
        self._tape = None
        if self.reducer is not None:
            for p in self.reducer.params:      # trainable parameters this loss does not reach: zero gradient, bucket complete
                if p not in grads:
                    self.reducer.grad_buffer(p).zero_()
                    self.reducer.grad_ready(p)
            self.reducer.finish()   # current stream waits for the (already overlapped) bucket all-reduces
        return grads

    def _as_nhwc(self, g, dtype=None):
        """Accept an NCHW-logical gradient (the autograd view) or an NHWC tensor; return dense NHWC in the engine dtype."""
        dtype = dtype or self.dtype
        if g.dim() != 4:
            raise EngineError("gradients must be 4-D")
        if g.stride(1) == 1 and g.stride(3) != 1:  # logical NCHW over NHWC memory
            g = g.permute(0, 2, 3, 1)
        elif g.stride(3) != 1:
            g = g.permute(0, 2, 3, 1).contiguous()
        if g.dtype != dtype:
            g = g.to(dtype)
        if not g.is_contiguous():
            g = g.contiguous()
        return g

    def _probe(self, name, g, raw, shortcut, gate=None, grads=None):
        """d loss / d (channel gate behind the conv `name`) = sum_{n,h,w} y * dL/d(y g) with y the conv's own output.  The
        stored raw tensor is the GATED output y g (the gate lives in the packed weights), plus the shortcut the epilogue added
        when the conv closes a residual block: sum (raw - shortcut) * dL/draw = g * dL/dg."""
        _, s2 = ops.channel_sums(g, a=raw)
        if shortcut is not None:
            _, s2b = ops.channel_sums(g, a=shortcut)
            s2 = s2 - s2b
        if gate is not None:
            s2 = s2 / gate.weight.detach().float()
            if gate.weight.requires_grad and grads is not None:
                gbuf = self._grad_like(gate.weight)
                gbuf.copy_(s2)
                grads[gate.weight] = gbuf
                self._grad_done(gate.weight)
        self.probe_grads[name] = s2

    # ---- parameter gradients ------------------------------------------------------------------------------------------
    def _conv_wgrad(self, conv, a_in, g, grads, cin=None, cin_rot=0, gate=None):
        """Weight gradient of a dense conv (a_in = its input as stored, g = gradient of its raw output).  cin_rot: a_in holds
        the conv's input channels rotated left by that many (see _w_fwd); the gradient is rotated back."""
        w = conv.weight
        if not w.requires_grad:
            return
        if cin is not None and cin != a_in.shape[3]:
            a_in = a_in[..., :cin]
        gw = self._grad_like(w)
        if cin_rot:
            tmp = torch.empty_like(gw)
            ops.conv2d_wgrad(a_in, g, tmp, conv.stride[0], conv.padding[0], conv.dilation[0])
            k = w.shape[1] - cin_rot
            gw[:, cin_rot:].copy_(tmp[:, :k])
            gw[:, :cin_rot].copy_(tmp[:, k:])
        else:
            ops.conv2d_wgrad(a_in, g, gw, conv.stride[0], conv.padding[0], conv.dilation[0])
        if gate is not None:   # g is the gradient of the gated output: d(y g)/dW = g[c] * dy/dW
            gw.mul_(gate.weight.detach().float().view(-1, 1, 1, 1))
        grads[w] = gw
        self._grad_done(w)

    def _wants_bn_sums(self, bn_seq, gate=None):
        bn = _bn_of(bn_seq)
        return bn.weight.requires_grad or bn.bias.requires_grad or (gate is not None and gate.weight.requires_grad)

    def _bn_param_grads(self, bn_seq, g_x, act, grads, sub=None, gate=None, sums=None):
        """Eval-mode BN weight/bias gradients.  g_x: gradient w.r.t. the BN input (already through the ReLU mask and the BN
        scale; `sub` = a tensor that was added to it afterwards, e.g. the shortcut gradient), act = relu(bn(x)).
        gate: a GateLayer behind the ReLU, folded into the epilogue (act = relu(bn(x)) * g, g_x carries scale * g): with
        S1 = sum g_x, S2 = sum g_x * act, d loss / d gate = S2 / (scale g^2), and the BN gradients take S2 / g for S2."""
        bn = _bn_of(bn_seq)
        want_gate = gate is not None and gate.weight.requires_grad
        if not (bn.weight.requires_grad or bn.bias.requires_grad or want_gate):
            return
        scale, _ = self._bn_fold(bn_seq)
        # sums: (S1, S2) already taken by the conv epilogue that produced g_x (ops.conv2d(bn_sums=...)) -- no second pass over it
        s1, s2 = sums if sums is not None else ops.channel_sums(g_x, sub=sub, a=act)
        if gate is not None:
            gv = gate.weight.detach().float()
            if want_gate:
                gbuf = self._grad_like(gate.weight)
                gbuf.copy_(torch.where(scale != 0, s2 / (scale * gv * gv), torch.zeros_like(s2)))
                grads[gate.weight] = gbuf
                self._grad_done(gate.weight)
            s2 = s2 / gv
        if bn.weight.requires_grad or bn.bias.requires_grad:
            self._bn_grads_from_sums(bn, scale, s1, s2, grads)

    def _bn_grads_from_sums(self, bn, scale, s1, s2, grads):
        dg, db = self._grad_like(bn.weight), self._grad_like(bn.bias)
        ops.bn_eval_param_grads(s1.contiguous(), s2.contiguous(), scale, bn.weight.detach().float().contiguous(),
                                bn.bias.detach().float().contiguous(), dg, db)
        for p, gbuf in ((bn.weight, dg), (bn.bias, db)):
            if p.requires_grad:
                grads[p] = gbuf
                self._grad_done(p)

    # ---- input gradients ------------------------------------------------------------------------------------------------
    def _dense_dgrad(self, site, g, in_hw=None, **ep):
        """Input gradient of a dense conv: a stride-1 conv with flipped taps; a strided conv's output gradient is first
        zero-inserted onto the input grid (mod4.block1: conv1 3x3/s2 and the 1x1/s2 projection)."""
        conv = site.mod
        N, H, W, _ = g.shape
        if site.stride != 1:
            if in_hw is None:
                raise EngineError(f"{site.name}: strided input gradient needs the input size")
            g = ops.zero_insert(g, site.stride, in_hw)
            H, W = in_hw
        out = self._new(N, H, W, conv.in_channels)
        ops.conv2d(g, self._w_dgrad(conv, gate=site.gate), 1, site.dil * (site.k - 1) - site.pad, site.dil, out_raw=out, **ep)
        return out

    def _cheap_bwd(self, site, a_in, mid, g, grads, need_in, defer=None, **ep):
        """Backward of dw -> pw: both weight gradients, and (optionally) the input gradient with epilogue `ep`.  With a
        `defer` list the depthwise input gradient is not run: (gradient of the depthwise output, flipped taps, site) is
        appended instead, for a caller that sums several of them in one launch (`ops.dwconv_sum`)."""
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
        if dw.weight.requires_grad and defer is None:
            gw = self._grad_like(dw.weight)
            ops.dwconv_wgrad(a_in, g_mid, gw, site.k, site.pad, site.dil)
            grads[dw.weight] = gw
            self._grad_done(dw.weight)
        if defer is not None:
            # the caller batches the branches that read one tensor: weight gradients in one launch (kd_dwconv_wgrad_multi),
            # input gradients summed in one launch (kd_dwconv_fwd_sum)
            defer.append((g_mid, self._w_dw(dw, True) if need_in else None, site, dw.weight if dw.weight.requires_grad else None))
            return None
        if not need_in:
            return None
        return ops.dwconv(g_mid, self._w_dw(dw, True), site.k, site.dil * (site.k - 1) - site.pad, site.dil, **ep)

    def _site_bwd(self, site, a_in, mid, g, grads, need_in, **ep):
        """Weight gradient(s) of one conv site and, when need_in, its input gradient through epilogue `ep`."""
        if site.cheap:
            return self._cheap_bwd(site, a_in, mid, g, grads, need_in, **ep)
        self._conv_wgrad(site.mod, a_in, g, grads, gate=site.gate)
        if not need_in:
            return None
        return self._dense_dgrad(site, g, in_hw=(a_in.shape[1], a_in.shape[2]), **ep)

    def _block_bwd(self, bi, rec, g_out, g_site_hint, grads):
        """g_out: gradient w.r.t. the block output (or None).  Returns the gradient w.r.t. the raw block input, or None when
        nothing upstream is trainable."""
        blk, sites = rec["blk"], rec["sites"]
        bns = {n: m for n, m in blk.convs.named_children() if n.startswith("bn")}
        g = g_out  # gradient w.r.t. the raw output of the current site (last site: the block output)
        for i in range(len(sites) - 1, -1, -1):
            site, a_in, need_in = sites[i], rec["a_in"][i], rec["rg_a"][i]
            g_in = None
            bn_seq = bns[f"bn{i + 1}"] if i > 0 else blk.bn1
            act_gate = _act_gate(bn_seq) if i > 0 else None     # a_in = relu(bn(c_{i-1})) * gate
            if g is not None and i in rec.get("probe", {}):
                self._probe(site.name, g, *rec["probe"][i], gate=site.gate, grads=grads)
            if g is not None and (need_in or site.trainable):
                sc, _ = self._act_fold(bn_seq, act_gate)
                ep = dict(mask=a_in, mask_scale=sc)
                sub = None
                if i > 0:
                    # a_in = relu(bn_{i+1}(c_{i-1})): mask epilogue gives d/dc_{i-1}; add that tensor's own hint gradient
                    sub = g_site_hint.get((bi, i - 1))
                    ep["res_post"] = sub
                elif g_out is not None:
                    if rec["proj"]:
                        psite = _Site(f"{rec['name']}.proj_conv", blk.proj_conv)
                        self._conv_wgrad(blk.proj_conv, a_in, g_out, grads)
                        # conv1 and proj_conv both read a_in: their input gradients as ONE K-concatenated 1x1 launch [g | g_out]
                        dual_bwd = (need_in and site.k == 1 and site.stride == 1 and g_out.is_contiguous() and g.is_contiguous() and
                                    self._dual_ok(g.shape, site.cout, g_out.shape[3], site.cin, 1, site.mod, blk.proj_conv, site))
                        if need_in and not dual_bwd:
                            ep["res_pre"] = self._dense_dgrad(psite, g_out, in_hw=(a_in.shape[1], a_in.shape[2]))
                    elif rec["rg_in"]:
                        sub = g_out
                        ep["res_post"] = g_out
                fused = []
                if _FUSE_BN_SUMS and need_in and not site.cheap and self._wants_bn_sums(bn_seq, act_gate):
                    ep["bn_sums"] = fused       # the dense input gradient's epilogue takes the BN parameter sums where it can
                if i == 0 and g_out is not None and rec["proj"] and dual_bwd:
                    self._conv_wgrad(site.mod, a_in, g, grads, gate=site.gate)
                    g_in = self._new(g.shape[0], g.shape[1], g.shape[2], site.cin)
                    try:
                        ops.conv2d(g, self._w_cat(site.mod, blk.proj_conv, "dgrad"), out_raw=g_in, x2=g_out, **ep)
                    except ops.DualUnsupported:      # refused on the real epilogue (nothing was launched): the two-launch form
                        if ep.get("bn_sums") is not None:
                            del ep["bn_sums"][:]
                        ep["res_pre"] = self._dense_dgrad(psite, g_out, in_hw=(a_in.shape[1], a_in.shape[2]))
                        g_in = self._dense_dgrad(site, g, in_hw=(a_in.shape[1], a_in.shape[2]), **ep)
                else:
                    g_in = self._site_bwd(site, a_in, rec["mid"][i], g, grads, need_in, **ep)
                if g_in is not None:
                    self._bn_param_grads(bn_seq, g_in, a_in, grads, sub=sub, gate=act_gate, sums=fused[0] if fused else None)
            elif i == 0 and g is None and g_out is not None:
                # only the shortcut carries gradient into this block's input
                if rec["proj"]:
                    self._conv_wgrad(blk.proj_conv, a_in, g_out, grads)
                    if need_in:
                        psite = _Site(f"{rec['name']}.proj_conv", blk.proj_conv)
                        sc, _ = self._bn_fold(blk.bn1)
                        g_in = self._dense_dgrad(psite, g_out, in_hw=(a_in.shape[1], a_in.shape[2]), mask=a_in, mask_scale=sc)
                        self._bn_param_grads(blk.bn1, g_in, a_in, grads)
                elif rec["rg_in"]:
                    g_in = g_out
            if i > 0:
                if g_in is None:
                    g_in = g_site_hint.get((bi, i - 1)) if rec["rg_a"][i] else None
                g = g_in
            else:
                if not rec["rg_in"]:
                    return None
                return g_in
        return None

    def _aspp_bwd(self, g_cat, g_cat_hint, g_aspp, grads):
        """g_cat: gradient w.r.t. the RAW branch outputs as one concat-shaped tensor (already through each branch's
        BN+ReLU, from the bot_aspp dgrad epilogue) or None; g_cat_hint: gradient w.r.t. the activated concat (an `aspp`
        hint); g_aspp: {branch: gradient w.r.t. that branch conv's raw output} (hints on aspp.features.N.0).
        Returns the gradient w.r.t. mod7's output, or None."""
        arec = self._tape["aspp"]
        aspp, cat, red, x7, rg7 = arec["mod"], arec["cat"], arec["red"], arec["x7"], arec["rg7"]
        lead = arec.get("lead", 1)
        cat_sums = arec.pop("cat_sums", None)     # (S1, S2) over the concat's channels from the bot_aspp input-gradient epilogue
        if g_cat_hint is not None:
            scale = self._cat_scale(aspp)
            g_cat = ops.relu_bn_bwd(g_cat_hint, cat, scale, res=g_cat)
            cat_sums = None                        # an `aspp` hint's gradient joined in: the sums no longer describe g_cat
        # x7 feeds every branch: its gradient is the sum of the branches' input gradients.  The replaced (cheap) branches go
        # first -- weight gradients per branch, their depthwise input gradients summed inside ONE launch per geometry
        # (registers instead of a read-modify-write of the running 4096-channel sum per branch) -- then the dense branches
        # chain onto that sum through their dgrad epilogue.
        g_x7 = None
        self._shape_grads = None
        if lead == 2 and g_cat is not None:
            self._edge_branch_bwd(arec, g_cat[..., red:2 * red], grads)
        todo = []
        for i, br in enumerate(arec["branches"]):
            sl = slice(red * (i + lead), red * (i + lead + 1))
            g = g_aspp.get(i)
            if g_cat is not None:
                gi = g_cat[..., sl]
                self._bn_param_grads(br["bn"], gi, cat[..., sl], grads, gate=br.get("act_gate"),
                                     sums=(cat_sums[0][sl], cat_sums[1][sl]) if cat_sums is not None else None)
                g = gi if g is None else g + gi
            if g is None:
                continue
            site = br["site"]
            if br.get("probe_raw") is not None:
                self._probe(site.name, g, br["probe_raw"], None, gate=site.gate, grads=grads)
            if site.cheap and not g.is_contiguous():
                g = g.contiguous()   # the depthwise / pointwise-wgrad kernels take dense views
            todo.append((site, br, g))
        deferred = []
        for site, br, g in todo:
            if site.cheap and _DW_SUM:
                self._cheap_bwd(site, x7, br["mid"], g, grads, rg7, defer=deferred)
            elif site.cheap:   # A/B: one launch per branch, the running sum chained through res_post
                g_in = self._cheap_bwd(site, x7, br["mid"], g, grads, rg7, res_post=g_x7)
                if g_in is not None:
                    g_x7 = g_in
        groups = {}
        for g_mid, w_t, site, w_dw in deferred:
            groups.setdefault((site.k, site.pad, site.dil, tuple(g_mid.shape), g_mid.dtype), []).append((g_mid, w_t, w_dw))
        for (k, pad, dil, _, _), members in groups.items():
            train = [(g_mid, w_dw) for g_mid, _, w_dw in members if w_dw is not None]
            if train:   # the branches' depthwise weight gradients: x7 staged once for all of them
                gws = [self._grad_like(w_dw) for _, w_dw in train]
                ops.dwconv_wgrad_multi(x7, [g for g, _ in train], gws, k, pad, dil)
                for (_, w_dw), gw in zip(train, gws):
                    grads[w_dw] = gw
                    self._grad_done(w_dw)
            items = [(g_mid, w_t) for g_mid, w_t, _ in members if w_t is not None]
            for j in range(0, len(items), 3):
                part = items[j:j + 3]
                if g_x7 is None:
                    g_x7 = ops.dwconv_sum([a for a, _ in part], [w for _, w in part], k, dil * (k - 1) - pad, dil)
                else:   # a second geometry / a fourth branch: chain onto the running sum
                    for a, w in part:
                        g_x7 = ops.dwconv(a, w, k, dil * (k - 1) - pad, dil, res_post=g_x7)
        for site, br, g in todo:
            if not site.cheap:
                g_in = self._site_bwd(site, x7, br["mid"], g, grads, rg7, res_post=g_x7)
                if g_in is not None:
                    g_x7 = g_in
        if g_cat is not None:
            g_x7 = self._image_pool_bwd(arec, g_cat[..., 0:red], grads, g_x7)
        return g_x7

    def _edge_branch_bwd(self, arec, gi, grads):
        """Edge branch of the GSCNN ASPP (gscnn.py:168-171: resampled edge attention -> 1x1 (1 -> red) -> BN -> ReLU) and, behind
        it, the whole shape stream.  gi: gradient w.r.t. the branch's conv output (already through BN + ReLU), a channel slice of
        the concat-shaped gradient.  Leaves {m1, m3, m4, m7: gradient | None} in self._shape_grads."""
        tape = self._tape
        if tape.get("shape") is None:
            raise EngineError("a gradient reached the GSCNN edge branch, but the shape stream was run without keeping its "
                              "intermediates (DepthwiseStudent.logits_need_grad / an `aspp` hint switch that on)")
        aspp, cat, red = arec["mod"], arec["cat"], arec["red"]
        conv, bn = aspp.edge_conv[0], aspp.edge_conv[1]
        acts = tape["acts"]
        N, H, W = acts.shape
        h8, w8 = cat.shape[1], cat.shape[2]
        self._bn_param_grads(bn, gi, cat[..., red:2 * red], grads)
        if conv.weight.requires_grad:
            e = ops.upsample_bilinear_ac(acts.unsqueeze(3), (h8, w8))                          # the branch's input (N,h8,w8,1) fp32
            pad8 = torch.zeros((N, h8, w8, 8), dtype=gi.dtype, device=gi.device)
            pad8[..., 0:1] = e.to(gi.dtype)
            dw8 = torch.empty((red, 8, 1, 1), dtype=torch.float32, device=gi.device)
            ops.pw_wgrad(pad8, gi, dw8)
            self._give(conv.weight, dw8[:, 0], grads)
        need = {"m7": arec["rg7"], "m1": self.net.mod1.conv1.weight.requires_grad}
        for nm in ("mod3", "mod4"):
            need["m" + nm[-1]] = bool(tape["mod_end"][nm][1])
        if not (any(need.values()) or any(p.requires_grad for m in self._shape_modules() for p in m.parameters())):
            return
        g_e = torch.empty((N, h8, w8, 1), dtype=gi.dtype, device=gi.device)
        ops.conv2d(gi, self._packed(conv.weight, ("edge_t", self.dtype),
                                    lambda: ops.pack_conv_weight(conv.weight.detach().float().view(1, red, 1, 1), self.dtype, KD_PACK_FWD)),
                   out_raw=g_e)
        g_acts = ops.upsample_bilinear_ac_bwd(g_e, (H, W), out_dtype=torch.float32).reshape(N, H, W)   # adjoint of the resample
        self._shape_grads = self._shape_stream_bwd(g_acts, grads, need)

    def _shape_modules(self):
        net = self.net
        return [net.dsn3, net.dsn4, net.dsn7, net.res1, net.res2, net.res3, net.d1, net.d2, net.d3, net.gate1, net.gate2, net.gate3,
                net.fuse, net.cw]

    def _cat_scale(self, aspp):
        """BN scales of [image branch, features 0..] concatenated: the mask_scale of the concat buffer."""
        bns = [(aspp.img_conv[1], None)] + ([(aspp.edge_conv[1], None)] if hasattr(aspp, "edge_conv") else []) + \
              [(br[1], _gate_split(br[2])[1] if len(br) > 2 else None) for br in aspp.features]
        key = tuple(id(_bn_of(b)) for b, _ in bns)
        vers = tuple(self._act_fold(b, gt)[0]._kd_serial for b, gt in bns)
        ent = self._bn.get(("cat", key))
        if ent is None or ent[0] != vers:
            ent = (vers, (torch.cat([self._act_fold(b, gt)[0] for b, gt in bns]).contiguous(), None), None)
            self._bn[("cat", key)] = ent
        return ent[1][0]

    def _image_pool_bwd(self, arec, g_img, grads, g_x7):
        """Backward of the image-pooling branch (deeplabv3.py:59-62,67-70): g_img is the gradient of the broadcast branch
        output, already masked and scaled per channel.  Per-image (N x 256 / N x 4096) algebra only; the two reductions
        over pixels and the broadcast are kernels."""
        aspp, x7, rg7, cat, red = arec["mod"], arec["x7"], arec["rg7"], arec["cat"], arec["red"]
        conv, bn = aspp.img_conv[0], _bn_of(aspp.img_conv[1])
        train = conv.weight.requires_grad or bn.weight.requires_grad or bn.bias.requires_grad
        if not (rg7 or train):
            return g_x7
        N, h8, w8, cin = x7.shape
        gz, _ = ops.channel_sums(g_img, per_image=True)             # (N, red): d loss / d conv output
        if train:
            mean, _ = ops.channel_sums(x7, per_image=True)
            mean = mean / float(h8 * w8)
            if conv.weight.requires_grad:
                gw = self._grad_like(conv.weight)
                gw.copy_((gz.t() @ mean).view_as(gw))
                grads[conv.weight] = gw
                self._grad_done(conv.weight)
            if bn.weight.requires_grad or bn.bias.requires_grad:
                y = cat[:, 0, 0, 0:red].float()                      # relu(bn(conv(mean))) per image
                scale, _ = self._bn_fold(aspp.img_conv[1])
                self._bn_grads_from_sums(bn, scale, gz.sum(0), (gz * y).sum(0), grads)
        if not rg7:
            return g_x7
        vec = (gz @ conv.weight.detach().float().view(red, cin)).contiguous()   # (N, cin)
        if g_x7 is None:
            g_x7 = self._new(N, h8, w8, cin)
            ops.broadcast_add(vec, g_x7, alpha=1.0 / float(h8 * w8), accumulate=False)
        else:
            ops.broadcast_add(vec, g_x7, alpha=1.0 / float(h8 * w8), accumulate=True)
        return g_x7

    def _decoder_bwd(self, g_logits, grads):
        """Backward of upsample / final / concat / bot_fine / upsample x4 / bot_aspp.  Returns (g_cat, g_m2): the gradient
        w.r.t. the raw ASPP branch outputs (concat-shaped, through BN+ReLU) and w.r.t. mod2's output -- None where nothing
        upstream is trainable."""
        net, dec = self.net, self._tape["dec"]
        if self.is_gscnn and self._tape.get("shape") is None:
            raise EngineError("a gradient reached the GSCNN logits, but the shape stream was run without keeping its intermediates: "
                              "set DepthwiseStudent.logits_need_grad = True before the forward (LayerwiseTrainer does for "
                              "trainer.backprop = 'kd+hint', TaylorPruneTrainer always); the shipped GSCNN plan back-propagates hint "
                              "losses only (cfg/cityscapes/51M_gscnn_all.json)")
        f = self._final()
        N, h2, w2, _ = dec["d2"].shape
        cdec, nf = dec["cdec"], dec["nf"]
        rg_dec0 = dec["rg_cat"] or dec["rg_m2"] or _is_trainable(net.bot_aspp) or _is_trainable(net.bot_fine)
        rg_d1 = rg_dec0 or _is_trainable(f[0]) or _is_trainable(f[1])
        rg_d2 = rg_d1 or _is_trainable(f[3]) or _is_trainable(f[4])
        if not (rg_d2 or _is_trainable(f[6])):
            return None, None
        ncls = f[6].out_channels
        kpad = ((ncls + 63) // 64) * 64
        g_d3 = self._new(N, h2, w2, kpad, zero=True)               # classes padded to the GEMM K granule
        ops.upsample_bilinear_ac_bwd(g_logits, (h2, w2), out=g_d3[..., :ncls], align_corners=not self.is_gscnn)   # gscnn.py:323
        if f[6].weight.requires_grad and self.dtype == torch.bfloat16 and kpad != ncls:
            # the classifier's weight gradient over the zero-padded class channels: 64 outputs take the LDS-DMA + transposing-read
            # kernel (19 do not: Cout % 8), the 45 extra rows are zeros
            tmp = torch.empty((kpad, f[6].weight.shape[1], 1, 1), dtype=torch.float32, device=g_d3.device)
            ops.conv2d_wgrad(dec["d2"], g_d3, tmp, 1, 0, 1)
            gw = self._grad_like(f[6].weight)
            gw.copy_(tmp[:ncls])
            grads[f[6].weight] = gw
            self._grad_done(f[6].weight)
        else:
            self._conv_wgrad(f[6], dec["d2"], g_d3[..., :ncls], grads)
        if not rg_d2:
            return None, None
        sc, _ = self._bn_fold(f[4])
        g_c2 = self._new(N, h2, w2, f[3].out_channels)
        fused = [] if _FUSE_BN_SUMS and self._wants_bn_sums(f[4]) else None
        ops.conv2d(g_d3, self._w_dgrad(f[6], cout_pad=kpad), out_raw=g_c2, mask=dec["d2"], mask_scale=sc, bn_sums=fused)
        self._bn_param_grads(f[4], g_c2, dec["d2"], grads, sums=fused[0] if fused else None)
        self._conv_wgrad(f[3], dec["d1"], g_c2, grads)
        if not rg_d1:
            return None, None
        sc, _ = self._bn_fold(f[1])
        fused = [] if _FUSE_BN_SUMS and self._wants_bn_sums(f[1]) else None
        g_c1 = self._dense_dgrad(_Site("final.3", f[3]), g_c2, mask=dec["d1"], mask_scale=sc, bn_sums=fused)
        self._bn_param_grads(f[1], g_c1, dec["d1"], grads, sums=fused[0] if fused else None)
        self._conv_wgrad(f[0], dec["dec0"], g_c1, grads, cin=cdec, cin_rot=nf)
        if not rg_dec0:
            return None, None
        nu = dec["nu"]
        kf = ((nf + 63) // 64) * 64
        g_dec0 = self._new(N, h2, w2, nu + kf)                     # [upsampled | fine | pad] like dec0
        if nu + kf > cdec:
            g_dec0[..., cdec:].zero_()
        wd = self._w_dgrad(f[0], cin_rot=nf)                       # (cdec, 3, 3, 256): rows = gradient channels of dec0
        if _SPLIT_DEC_DGRAD and self.dtype == torch.bfloat16 and nu % 256 == 0 and cdec > nu:
            # 304 output channels take the one-tile-per-workgroup row kernel with a ragged second N tile (0.73 PFLOP/s); the
            # 256 upsampled channels alone fit the persistent kernel, the 48 fine channels a narrow tile
            ops.conv2d(g_c1, wd[:nu], 1, 1, 1, out_raw=g_dec0[..., 0:nu])
            ops.conv2d(g_c1, wd[nu:cdec], 1, 1, 1, out_raw=g_dec0[..., nu:cdec])
        else:
            ops.conv2d(g_c1, wd, 1, 1, 1, out_raw=g_dec0[..., 0:cdec])
        # bot_fine (1x1 on mod2's output)
        g_m2 = None
        self._conv_wgrad(net.bot_fine, dec["m2"], g_dec0[..., nu:cdec], grads)
        if dec["rg_m2"]:
            g_m2 = self._new(N, h2, w2, net.bot_fine.in_channels)
            # K = 48 -> 64: the 16 extra input channels are the zeroed pad and meet zero weight rows
            ops.conv2d(g_dec0[..., nu:nu + kf], self._w_dgrad(net.bot_fine, cout_pad=kf), out_raw=g_m2)
        # upsample x4 and bot_aspp (1x1 on the ASPP concat)
        g_cat = None
        if dec["rg_cat"] or _is_trainable(net.bot_aspp):
            g_up = ops.upsample_bilinear_ac_bwd(g_dec0[..., 0:nu], dec["small"])
            self._conv_wgrad(net.bot_aspp, dec["cat"], g_up, grads)
            if dec["rg_cat"]:
                cat = dec["cat"]
                g_cat = self._new(N, cat.shape[1], cat.shape[2], cat.shape[3])
                # the branches' BN parameter sums are channel slices of this epilogue's sums (kept for _aspp_bwd)
                fused = [] if _FUSE_BN_SUMS else None
                ops.conv2d(g_up, self._w_dgrad(net.bot_aspp), out_raw=g_cat, mask=cat, mask_scale=self._cat_scale(net.aspp), bn_sums=fused)
                self._tape["aspp"]["cat_sums"] = fused[0] if fused else None
        return g_cat, g_m2


class _StudentFunction(torch.autograd.Function):
    """One autograd node for the whole student: (x, *trainable params) -> (logits, *hints)."""

    @staticmethod
    def forward(ctx, engine, x, *params):
        ctx.set_materialize_grads(False)
        logits, hints = engine.forward(x, prefix=engine._next_prefix)
        ctx.engine, ctx.params = engine, params
        lazy = isinstance(logits, LazyLogits)
        outs = (logits if lazy else logits.permute(0, 3, 1, 2),) + tuple(h.permute(0, 3, 1, 2) for h in hints)
        ctx.nh = len(hints)
        if lazy:
            # the logits output carries no gradient of its own; a 0-dim differentiable output anchors loss values computed from
            # the half-resolution logits to this node (lazy.deferred: if one is back-propagated, its gradient arrives through it)
            ctx.mark_non_differentiable(outs[0])
            outs = outs + (x.new_zeros(()),)
        ctx.lazy = outs[0] if lazy else None
        return outs

    @staticmethod
    def backward(ctx, g_logits, *g_rest):
        g_hints = g_rest[:ctx.nh]
        if ctx.lazy is not None:      # a loss on the lazy logits was back-propagated after all: lazy._DeferredLogitLoss left its gradient
            g_logits, ctx.lazy.pending_grad = ctx.lazy.pending_grad, None
            ctx.lazy = None
        grads = ctx.engine.backward(list(g_hints), g_logits)
        return (None, None) + tuple(grads.get(p) for p in ctx.params)


def run_student(engine, x, prefix=None):
    """Differentiable student call: returns (logits NCHW-logical fp32, [hints NCHW-logical]).  prefix: the teacher engine's
    export of the shared frozen prefix (StudentEngine.shareable_prefix), or None."""
    engine._next_prefix = prefix
    engine._lazy_call = True
    try:
        return _run_student(engine, x, prefix)
    finally:
        engine._next_prefix = None
        engine._lazy_call = False


def _run_student(engine, x, prefix):
    params = tuple(p for p in engine.net.parameters() if p.requires_grad)
    if torch.is_grad_enabled() and (params or engine.probe_names):
        if not params:   # probes only: autograd still needs one differentiable input to build the node
            x = x.detach().requires_grad_(True)
        engine._probe_active = bool(engine.probe_names)
        engine._differentiable = True
        try:
            outs = _StudentFunction.apply(engine, x, *params)
            if isinstance(outs[0], LazyLogits):
                outs[0].anchor = outs[-1]
                outs = outs[:-1]
        finally:
            engine._probe_active = False
            engine._differentiable = False
    else:
        logits, hints = engine.forward(x, prefix=prefix)
        engine._tape = None
        outs = (logits if isinstance(logits, LazyLogits) else logits.permute(0, 3, 1, 2),) + tuple(h.permute(0, 3, 1, 2) for h in hints)
    return outs[0], list(outs[1:])
