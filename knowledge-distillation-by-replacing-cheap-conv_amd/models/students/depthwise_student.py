"""`DepthwiseStudent`: frozen teacher + student copy whose named blocks are swapped for cheap convs.

API mirror of the reference's models/students/depthwise_student.py (replace :86-124, register_hint_layers
:46-78, unfreeze :80-84, get_block :151-166, _set_block :134-149, forward :168-177, inference :179-185,
reset :261-273, train :275-283).  What differs is how forward executes:
  * a DeepWV3Plus student runs through engine.StudentEngine (hand-written HIP kernels, hints captured by name);
  * the teacher stays a PyTorch-ROCm module under no_grad, in channels_last at the engine's dtype, on a side
    HIP stream so its kernels overlap the student's (its logits / hints are only needed by the losses);
  * any other architecture (e.g. the CIFAR ResNet-20 plumbing config) falls back to the reference's own
    mechanism -- module forward + hooks -- with the replaced blocks still HIP-backed.
"""
import copy
import gc
from functools import reduce

import torch
from torch import nn

from ..deeplabv3 import DeepWV3Plus
from ..gscnn import GSCNN
from ...lazy import LazyLogits
from .transform_blocks import DepthwiseSeparableBlock

BLOCKS_LEVEL_SPLIT_CHAR = '.'


class DepthwiseStudent(nn.Module):
    def __init__(self, teacher_model, config=None, dtype=None):
        super().__init__()
        self.config = config
        self.teacher = copy.deepcopy(teacher_model)
        if self.teacher.training:
            self.teacher.eval()
        for param in self.teacher.parameters():
            param.requires_grad = False
        self.student = copy.deepcopy(self.teacher)  # eval mode and frozen, like its source

        self.replaced_block_names = []
        self.student_hidden_outputs = list()
        self.teacher_hidden_outputs = list()
        self._student_hook_handlers = list()
        self._teacher_hook_handlers = list()
        self.aux_block_names = list()
        self.hint_block_names = list()
        self.student_hint_names = list()   # names of student_hidden_outputs, forward-execution order (fused path)
        self.save_hidden = True

        # compute dtype of the fused path: bf16 (measured path) unless the config / caller asks for fp32 parity mode
        if dtype is None:
            dtype = torch.bfloat16
            try:
                if config is not None and str(config['trainer'].get('dtype', 'bf16')) in ('fp32', 'float32', 'f32'):
                    dtype = torch.float32
            except (KeyError, TypeError, AttributeError):
                pass
        self.dtype = dtype
        self._engine = None
        self._teacher_ready = None
        self._side_stream = None
        self.overlap_teacher = True
        # "hip":   the frozen teacher graph runs through the engine's kernels (fused BN/ReLU/add/concat); default for
        #          DeepWV3Plus: 45 % of the step's FLOPs are the teacher's, and it avoids MIOpen's first-run solver search.
        # "torch": the teacher runs as a PyTorch-ROCm module (MIOpen convs) on a side stream (north_star's split).
        self.teacher_backend = "hip"
        self.hip_teacher_side_stream = False
        # Opt-in: compute the frozen layers the student shares bit for bit with the teacher (everything before the first
        # replaced / unfrozen block) once per step instead of once per network.  Same numbers, fewer FLOPs than the
        # reference's two full forwards (which is why it is not the default and not what bench.py's headline measures).
        self.share_frozen_prefix = False
        self._teacher_engine = None
        # Gated-SCNN only: a loss term will be back-propagated into the student LOGITS (KD / supervised terms; the shipped plan's
        # hint losses never reach them).  The shape stream then keeps its intermediates and runs on the differentiable kernels.
        # Trainers set it from their loss (LayerwiseTrainer: trainer.backprop = 'kd+hint'; TaylorPruneTrainer: always).
        self.logits_need_grad = False

    # ------------------------------------------------------------------ model surgery (host side)
    def register_hint_layers(self, block_names):
        if self.fused and len(block_names) > 0:
            from ...engine import StudentEngine
            StudentEngine(self.student, self.dtype).check_hint_names(block_names)   # fail at plan time, not mid-run
        if len(block_names) > 0:
            self._remove_hooks()
            self.hint_block_names = []
        for block_name in block_names:
            self.aux_block_names.append(block_name)
            self.hint_block_names.append(block_name)
            teacher_block = self.get_block(block_name, self.teacher)
            student_block = self.get_block(block_name, self.student)

            def teacher_handle(m, inp, out):
                if self.save_hidden:
                    self.teacher_hidden_outputs.append(out)

            self._teacher_hook_handlers.append(teacher_block.register_forward_hook(teacher_handle))

            def student_handle(m, inp, out):
                if self.save_hidden:
                    self.student_hidden_outputs.append(out)

            # only reached on the non-fused path; the engine captures student hints by name
            self._student_hook_handlers.append(student_block.register_forward_hook(student_handle))
        gc.collect()

    def unfreeze(self, block_names):
        for block_name in block_names:
            block = self.get_block(block_name, self.student)
            for param in block.parameters():
                param.requires_grad = True

    def replace(self, blocks, **kwargs):
        """blocks: [{"name": 'mod4.block2.convs.conv2', "epoch": 1, "args"(optional): {kernel_size, padding, dilation}}]"""
        for block in blocks:
            block_name = block['name']
            self.replaced_block_names.append(block_name)
            teacher_block = self.get_block(block_name, self.teacher)
            args = block['args'] if "args" in block else kwargs
            replace_block = DepthwiseSeparableBlock(in_channels=teacher_block.in_channels,
                                                    out_channels=teacher_block.out_channels,
                                                    kernel_size=args['kernel_size'],
                                                    padding=args['padding'],
                                                    dilation=args['dilation'],
                                                    groups=teacher_block.in_channels,
                                                    bias=teacher_block.bias)
            ref = next(self.student.parameters())
            replace_block.to(ref.device)
            self._set_block(block_name, replace_block, self.student)
        if self._engine is not None:
            self._engine.drop_caches()   # the freed dense weights' cache entries must not outlive them
        gc.collect()

    def _remove_hooks(self):
        while self._student_hook_handlers:
            self._student_hook_handlers.pop().remove()
        while self._teacher_hook_handlers:
            self._teacher_hook_handlers.pop().remove()

    def _set_block(self, block_name, block, model):
        parts = block_name.split(BLOCKS_LEVEL_SPLIT_CHAR)
        if len(parts) == 1:
            setattr(model, block_name, block)
        else:
            obj = self.get_block(BLOCKS_LEVEL_SPLIT_CHAR.join(parts[:-1]), model)
            if parts[-1].isdigit():
                obj[int(parts[-1])] = block
            else:
                setattr(obj, parts[-1], block)

    def get_block(self, block_name, model):
        def step(acc, elem):
            return acc[int(elem)] if elem.isdigit() else getattr(acc, elem)
        return reduce(step, block_name.split(BLOCKS_LEVEL_SPLIT_CHAR), model)

    # ------------------------------------------------------------------ execution
    @property
    def fused(self):
        return isinstance(self.student, (DeepWV3Plus, GSCNN))

    def _student_engine(self):
        from ...engine import StudentEngine
        if self._engine is None or self._engine.net is not self.student or self._engine.dtype != self.dtype:
            self._engine = StudentEngine(self.student, self.dtype)
        self._engine.hint_names = list(self.hint_block_names) if self.save_hidden else []
        self._engine.logits_need_grad = bool(getattr(self, "logits_need_grad", False))
        return self._engine

    def _prepare_teacher(self, device):
        """Teacher in channels_last at the engine dtype (weights converted once per device/dtype)."""
        key = (device, self.dtype)
        if self._teacher_ready != key:
            self.teacher.to(device=device, dtype=self.dtype, memory_format=torch.channels_last)
            self._teacher_ready = key

    def _teacher_forward(self, x):
        if self.teacher_backend == "hip" and isinstance(self.teacher, (DeepWV3Plus, GSCNN)):
            from ...engine import StudentEngine
            if self._teacher_engine is None or self._teacher_engine.net is not self.teacher or \
                    self._teacher_engine.dtype != self.dtype:
                self._teacher_engine = StudentEngine(self.teacher, self.dtype)
            eng = self._teacher_engine
            eng.edge_prior = getattr(self, "_shared_prior", None) if eng.is_gscnn else None
            eng.hint_names = list(self.hint_block_names) if self.save_hidden else []
            k = 0
            if self.share_frozen_prefix and self.fused and not self.hip_teacher_side_stream:
                k = self._student_engine().shareable_prefix(eng)
            eng._lazy_call = True          # the trainer's criteria read the half-resolution logits (lazy.LazyLogits)
            try:
                with torch.no_grad():
                    logits, hints = eng.forward(x, export_at=k if k > 0 else None)
            finally:
                eng._lazy_call = False
            self._prefix = eng.exported if k > 0 else None
            eng.exported = None
            eng._tape = None
            if self.save_hidden:
                self.teacher_hidden_outputs = [h.permute(0, 3, 1, 2) for h in hints]
            return logits if hasattr(logits, "materialize") else logits.permute(0, 3, 1, 2)     # (lazy.LazyLogits is NCHW-logical already)
        self._prepare_teacher(x.device)
        xt = x.to(dtype=self.dtype, memory_format=torch.channels_last)
        with torch.no_grad():
            return self.teacher(xt)

    def prefetch_teacher(self, x):
        """Start the frozen teacher's forward for the NEXT batch `x` now, on the side HIP stream: called between the criteria and
        `loss.backward()` of the current step, the teacher's logits / hints for the next step are produced while the student's
        backward runs on the main stream (the placement BASELINE's north_star describes; the reference runs the teacher inside
        `forward`, models/students/depthwise_student.py:168-177).  The next `forward(x)` -- the same tensor object, unmodified --
        takes these outputs instead of running the teacher; any other input discards them.  Same values bit for bit (the teacher is
        frozen and deterministic).  Returns False where it does not apply (non-fused students, Gated-SCNN's shared Canny prior,
        `share_frozen_prefix`)."""
        self._prefetched = None
        if not self.fused or not x.is_cuda or self.share_frozen_prefix or isinstance(self.teacher, GSCNN):
            return False
        if self._side_stream is None or self._side_stream.device != x.device:
            self._side_stream = torch.cuda.Stream(device=x.device)
        main = torch.cuda.current_stream()
        ready = torch.cuda.Event()
        ready.record(main)                       # x has been produced / transferred on the main stream
        self._side_stream.wait_event(ready)      # ... and nothing else of the main stream is waited for: the backward runs beside it
        x.record_stream(self._side_stream)       # (the batch may be dropped by its owner before the side stream has read it)
        keep = self.teacher_hidden_outputs       # (the current step's hints stay what the criteria saw)
        # a list of its own for the prefetched forward: the engine teacher rebinds the attribute, but a hooked PyTorch teacher
        # (teacher_backend "torch") APPENDS to whatever list is there -- into `keep`, it would pair the next step's student hints
        # with this step's teacher hints
        self.teacher_hidden_outputs = []
        # experiment: KDCC_TEACHER_CUS=n caps the persistent conv grids of the prefetched teacher's launches,
        # KDCC_STUDENT_CUS=m those of everything launched afterwards -- the two networks then share the chip by CUs instead of by time
        import os
        tcus, scus = os.environ.get("KDCC_TEACHER_CUS"), os.environ.get("KDCC_STUDENT_CUS")
        if tcus is not None:
            from ... import _lib
            _lib.check(_lib.lib().kd_conv_set_persist_cus(int(tcus)), "kd_conv_set_persist_cus")
        with torch.cuda.stream(self._side_stream):
            pred = self._teacher_forward(x)
        if tcus is not None:
            _lib.check(_lib.lib().kd_conv_set_persist_cus(int(scus or 0)), "kd_conv_set_persist_cus")
        hints = self.teacher_hidden_outputs
        self.teacher_hidden_outputs = keep
        self._prefix = None
        done = torch.cuda.Event()
        done.record(self._side_stream)
        self._prefetched = ((x.data_ptr(), tuple(x.shape), x._version), pred, hints, done)
        return True

    def forward(self, x):
        self.student_hidden_outputs = []
        self.teacher_hidden_outputs = []
        pre, self._prefetched = getattr(self, "_prefetched", None), None
        if not self.fused:
            with torch.no_grad():
                teacher_pred = self.teacher(x)
            return self.student(x), teacher_pred
        if not x.is_cuda:
            raise RuntimeError("the fused DeepWV3Plus student runs on the GPU only (no CPU fallback)")
        from ...engine import run_student
        engine = self._student_engine()
        if engine.is_gscnn:   # one Canny prior per batch, shared by teacher and student (the reference computes it twice, on the host)
            engine.edge_prior = engine.compute_edge_prior(x)
            if self._teacher_engine is not None:
                self._teacher_engine.edge_prior = engine.edge_prior
            self._shared_prior = engine.edge_prior
        hip_teacher = self.teacher_backend == "hip" and isinstance(self.teacher, (DeepWV3Plus, GSCNN))
        if pre is not None and pre[0] == (x.data_ptr(), tuple(x.shape), x._version):
            # the teacher already ran for this batch, on the side stream under the previous step's backward (prefetch_teacher)
            _, teacher_pred, thints, done = pre
            main = torch.cuda.current_stream()
            main.wait_event(done)
            for t in [teacher_pred] + list(thints):
                if isinstance(t, LazyLogits) and t.pending:
                    t.low.record_stream(main)
                else:
                    t.record_stream(main)
            self.teacher_hidden_outputs = thints
            self.prefetch_hits = getattr(self, "prefetch_hits", 0) + 1
            self._prefix = None
            student_pred, hints = run_student(engine, x)
        elif self.overlap_teacher and (not hip_teacher or self.hip_teacher_side_stream):
            # frozen teacher on a side stream: its logits/hints are consumed only by the losses, so it can overlap the
            # student's forward on the main stream (PyTorch-ROCm teacher: always; engine teacher: opt-in, since two
            # chip-filling MFMA kernel streams only trade tail effects -- measured in DESIGN.md section 5)
            if self._side_stream is None or self._side_stream.device != x.device:
                self._side_stream = torch.cuda.Stream(device=x.device)
            main = torch.cuda.current_stream()
            self._side_stream.wait_stream(main)
            with torch.cuda.stream(self._side_stream):
                teacher_pred = self._teacher_forward(x)
            student_pred, hints = run_student(engine, x)
            main.wait_stream(self._side_stream)
            for t in [teacher_pred] + list(self.teacher_hidden_outputs):
                if isinstance(t, LazyLogits) and t.pending:     # (record_stream on the wrapper would materialise the full-resolution tensor)
                    t.low.record_stream(main)
                else:
                    t.record_stream(main)
        else:
            self._prefix = None
            teacher_pred = self._teacher_forward(x)
            student_pred, hints = run_student(engine, x, prefix=self._prefix)
            self._prefix = None
        if self.save_hidden:
            self.student_hidden_outputs = hints
            self.student_hint_names = list(engine.last_hint_names)
        engine.edge_prior = None
        self._shared_prior = None
        if self._teacher_engine is not None:
            self._teacher_engine.edge_prior = None
        return student_pred, teacher_pred

    def inference(self, x):
        self.student_hidden_outputs = []
        self.teacher_hidden_outputs = []
        if not self.fused:
            return self.student(x)
        from ...engine import run_student
        engine = self._student_engine()
        engine.hint_names = []
        with torch.no_grad():
            pred, _ = run_student(engine, x)
        return pred

    @staticmethod
    def sliding_windows(h, w, tile, overlap=1 / 3):
        """Window boxes (x1, y1, x2, y2) of the reference's tiling (utils/tta_process.py:68-101), in its order (x outer, y
        inner): square `tile`-sized windows, stride ceil(tile * (1 - overlap)), the last window of a row / column pulled back
        inside the image, windows clipped to the image when it is smaller than the tile."""
        from math import ceil
        stride = ceil(tile * (1 - overlap))
        n_x = max(int(ceil((w - tile) / stride) + 1), 1)
        n_y = max(int(ceil((h - tile) / stride) + 1), 1)
        boxes = []
        for ix in range(n_x):
            for iy in range(n_y):
                x2, y2 = min(ix * stride + tile, w), min(iy * stride + tile, h)
                boxes.append((max(x2 - tile, 0), max(y2 - tile, 0), x2, y2))
        return boxes

    def inference_test(self, data, args, max_windows_per_pass=8):
        """Multi-scale sliding-window + horizontal-flip test-time inference of the student (reference depthwise_student.py:187-206
        with utils/tta_process.py), entirely on the device.  Per scale: the image and its mirror image are cut into the
        reference's windows (tile = int(scale * crop_size)), the windows go through the student in batches, their logits are
        summed into full-frame maps and divided by the window count, the mirrored map is flipped back, both are resized to the
        original size and averaged; the scales are averaged.  data: normalised (N,3,H,W) batch, as the trainer feeds it.

        args: {scales, crop_size[, window_count]}.  window_count = "reference" (default) reproduces the reference's count array
        bit for bit: it is allocated (C,h,w) and indexed `count[y1:y2, x1:x2] += 1` (tta_process.py:41-48), i.e. classes y1..y2
        and rows x1..x2 of ALL columns -- for the shipped 1024x2048 / crop 1024 case a per-row factor that leaves the arg-max
        alone; pinned against the reference's own reverse_mapping (tests/golden/tta.npz).  window_count = "pixel" divides by the
        true per-pixel number of windows instead.

        Resampling (scales != 1.0) is done on the normalised float tensor with bilinear interpolation at half-pixel centres
        (anti-aliased when shrinking, like PIL's BILINEAR); the reference resamples the uint8 PIL image and cv2-resizes the
        logits -- same geometry, not the same rounding, so scaled passes are close to, not bit-identical with, the reference.
        Every shipped config uses scales = [1.0], which involves no resampling at all."""
        import torch.nn.functional as F
        self.student_hidden_outputs = []
        self.teacher_hidden_outputs = []
        scales = [float(sc) for sc in args.get('scales', [1.0])]
        crop = int(args['crop_size'])
        mode = str(args.get('window_count', 'reference'))
        if mode not in ('reference', 'pixel'):
            raise ValueError("test.args.window_count must be 'reference' or 'pixel'")
        N, _, H, W = data.shape
        outs = []
        with torch.no_grad():
            for n in range(N):
                img = data[n:n + 1]
                per_scale = []
                for scale in scales:
                    if abs(scale - 1.0) < 1e-9:
                        img_s = img
                    else:
                        img_s = F.interpolate(img.float(), size=(int(H * scale), int(W * scale)), mode='bilinear', align_corners=False,
                                              antialias=scale < 1.0).to(img.dtype)
                    hs, ws = img_s.shape[2:]
                    boxes = self.sliding_windows(hs, ws, int(scale * crop))
                    acc = None
                    for flipped in (False, True):
                        src = torch.flip(img_s, dims=[3]) if flipped else img_s
                        full = cnt = None
                        for i in range(0, len(boxes), max_windows_per_pass):
                            part = boxes[i:i + max_windows_per_pass]
                            wins = torch.cat([src[:, :, y1:y2, x1:x2] for (x1, y1, x2, y2) in part], 0).contiguous()
                            logits = self.inference(wins).float()
                            if full is None:
                                C = logits.shape[1]
                                full = torch.zeros((C, hs, ws), dtype=torch.float32, device=data.device)
                                cnt = torch.zeros((C, hs, ws) if mode == 'reference' else (1, hs, ws), dtype=torch.float32,
                                                  device=data.device)
                            for j, (x1, y1, x2, y2) in enumerate(part):
                                full[:, y1:y2, x1:x2] += logits[j][:, :y2 - y1, :x2 - x1]
                                if mode == 'reference':
                                    cnt[y1:y2, x1:x2] += 1       # the reference's indexing: [class, row] of a (C,h,w) array
                                else:
                                    cnt[:, y1:y2, x1:x2] += 1
                        full = full / cnt
                        if flipped:
                            full = torch.flip(full, dims=[2])
                        if (hs, ws) != (H, W):
                            full = F.interpolate(full.unsqueeze(0), size=(H, W), mode='bilinear', align_corners=False)[0]
                        acc = full if acc is None else (acc + full) / 2
                    per_scale.append(acc)
                outs.append(torch.stack(per_scale).mean(0).unsqueeze(0))
        return torch.cat(outs, 0)

    # ------------------------------------------------------------------ bookkeeping
    def dump_trainable_params(self):
        params = sum(p.numel() for p in self.parameters() if p.requires_grad)
        return '\nTrainable parameters: {}'.format(params)

    def dump_student_teacher_blocks_info(self):
        rows = ["Block name | old block params | new block params"]
        for name in self.replaced_block_names:
            t = sum(p.numel() for p in self.get_block(name, self.teacher).parameters())
            s = sum(p.numel() for p in self.get_block(name, self.student).parameters())
            rows.append(f"{name} | {t} | {s}")
        return "\n".join(rows)

    def __str__(self):
        return super().__str__() + '\n' + self.dump_student_teacher_blocks_info()

    def reset(self):
        self._remove_hooks()
        self.hint_block_names = []
        while self.replaced_block_names:
            block_name = self.replaced_block_names.pop()
            teacher_block = self.get_block(block_name, self.teacher)
            self._set_block(block_name, copy.deepcopy(teacher_block).float(), self.student)
        self._engine = None

    def train(self, mode=True):
        self.save_hidden = bool(mode)
        super().train(mode)
        self.teacher.eval()  # teacher will always be in eval mode
        return self
