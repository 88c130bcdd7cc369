"""`TaylorPruneStudent`: first-order Taylor importance of the filters behind named blocks (reference
models/students/taylor_prune_student.py:9-66, transform_blocks/gate.py:5-12): `replace` swaps the named student block for
Sequential(copy of the teacher's block, GateLayer) -- a trainable per-channel multiplier initialised to 1 -- and
`get_gate_importance` reads (gate * d loss / d gate)^2 after loss.backward() (trainer/taylor_prune_trainer.py:204-211).

Same module surgery, parameters and state-dict keys as the reference (`<block>.0.*` = the copied block, `<block>.1.weight` =
the gate), so the gates are student parameters: the trainer's optimizer holds and steps them, and `.grad` accumulates across
backward calls exactly as autograd does there.  Execution differs: the fused engine never runs a gate as a kernel.  A gate
behind a conv is folded into that conv's packed weights (forward and input gradient), a gate behind a BN+ReLU into the
epilogue's scale / shift (positive gates: relu(y) * g == relu(y * g)); the gate gradients are per-channel reductions the
engine performs during backward (sum y * dL/dy / g behind a conv; the BN-parameter channel sums behind a ReLU).

Gate sites the fused graph supports -- every class cfg/taylor_importance_track.json uses: a block's conv
(`modN.blockM.convs.convK`), the ReLU of a bnrelu inside a block (`modN.blockM.convs.bnK.1`), an ASPP branch's conv or ReLU
(`aspp.features.N.0`, `aspp.features.N.2`).  Anything else is refused when the plan is applied."""
import copy
import re

import torch
from torch import nn

from .depthwise_student import DepthwiseStudent
from .transform_blocks import GateLayer  # noqa: F401  (re-exported: reference code imports it from here)

_CONV_SITE = re.compile(r"^(mod[2-7]\.block\d+\.convs\.conv\d+|aspp\.features\.\d+\.0)$")
_RELU_SITE = re.compile(r"^(mod[2-7]\.block\d+\.convs\.bn\d+\.1|aspp\.features\.\d+\.2)$")


class TaylorPruneStudent(DepthwiseStudent):
    def __init__(self, teacher_model, config=None, dtype=None):
        super().__init__(teacher_model, config, dtype=dtype)
        self.added_gates = dict()
        # this student exists to back-propagate the supervised loss through the logits (trainer/taylor_prune_trainer.py:204-206):
        # the engine hands out real (not lazy) logits
        self.logits_need_grad = True

    def _channels_behind(self, name, block):
        """Channel count of the tensor the gate multiplies; raises for a site the fused graph cannot fold a gate into."""
        if self.fused:
            if _CONV_SITE.match(name) and isinstance(block, nn.Conv2d):
                return block.out_channels
            if _RELU_SITE.match(name) and isinstance(block, nn.ReLU):
                parent = self.get_block(name.rsplit('.', 1)[0], self.teacher)
                return parent[int(name.rsplit('.', 1)[1]) - 1].num_features     # the BatchNorm2d in front of the ReLU
            raise ValueError(f"{name}: the fused student graph folds gates behind block convs (`...convs.convK`), bnrelu ReLUs "
                             f"(`...convs.bnK.1`) and ASPP branch convs / ReLUs (`aspp.features.N.0|2`) only")
        return None

    def replace(self, blocks, **kwargs):
        """blocks: [{"name": block, "epoch": e, "num_features": C}] -- the named student block becomes
        Sequential(copy of the teacher's block, GateLayer(C)) (reference :21-43)."""
        for block in blocks:
            name, num_features = block['name'], block['num_features']
            teacher_block = self.get_block(name, self.teacher)
            channels = self._channels_behind(name, teacher_block)
            if channels is not None and channels != num_features:
                raise ValueError(f"{name}: num_features {num_features} != {channels} channels behind the block")
            self.replaced_block_names.append(name)
            ref = next(self.student.parameters())
            gate = GateLayer(num_features).to(ref.device)
            self.added_gates[name] = gate
            # (the copied block keeps the frozen teacher's parameters: requires_grad False, like the reference)
            self._set_block(name, nn.Sequential(copy.deepcopy(teacher_block).float().to(ref.device), gate), self.student)
        if self._engine is not None:
            self._engine.drop_caches()

    def reset(self):
        super().reset()
        self.added_gates = dict()

    def get_gate_importance(self):
        """{gate name: (gate * gate.grad)^2 as a numpy vector} (reference :58-66); gate.grad is whatever autograd has
        accumulated since the optimizer last zeroed it."""
        out = {}
        for name, gate in self.added_gates.items():
            if gate.weight.grad is None:
                raise RuntimeError(f"no gradient reached the gate behind {name}: call loss.backward() on a loss of the student logits first")
            out[name] = ((gate.weight.detach() * gate.weight.grad) ** 2).detach().cpu().numpy()
        return out
