"""`TaylorPruneStudent`: first-order Taylor importance of the filters of named convs (reference
models/students/taylor_prune_student.py:9-66, transform_blocks/gate.py:5-12): the reference wraps a copy of the teacher block
in Sequential(block, GateLayer) -- a per-channel multiplier initialised to 1 -- and reads importance = (gate * d loss/d gate)^2
after loss.backward() (trainer/taylor_prune_trainer.py:204-211).

Here the student graph is not edited at all: a unit gate does not change the forward, and its gradient
d loss/d gate[c] = sum_{n,h,w} y[n,c,h,w] * dL/dy[n,c,h,w] is one per-channel reduction the engine performs during backward
for the conv sites registered as probes (engine.StudentEngine.probe_names -> kd_channel_sums).  The GateLayer objects exist
for API / bookkeeping parity (`added_gates`, `num_features`, `.weight`, `.weight.grad`).  Deviation, on purpose: the gates
are not handed to the optimizer, so they stay exactly 1 (the reference's optimizer nudges them every step, which perturbs the
very network whose filters are being ranked).  Gates can be placed behind convs (block conv sites, ASPP branch convs)."""
import torch
from torch import nn

from .depthwise_student import DepthwiseStudent


class GateLayer(nn.Module):
    def __init__(self, num_features):
        super().__init__()
        self.num_features = num_features
        self.weight = nn.Parameter(torch.ones(num_features))

    def forward(self, input):
        return input * self.weight.view(1, -1, 1, 1)


class TaylorPruneStudent(DepthwiseStudent):
    def __init__(self, teacher_model, config=None, dtype=None):
        super().__init__(teacher_model, config, dtype=dtype)
        self.added_gates = dict()

    def replace(self, blocks, **kwargs):
        """blocks: [{"name": conv site, "epoch": e, "num_features": C}] -- registers a gate (probe) behind each named conv."""
        for block in blocks:
            name = block['name']
            conv = self.get_block(name, self.teacher)
            if not isinstance(conv, nn.Conv2d) or conv.out_channels != block['num_features']:
                raise ValueError(f"{name}: a gate needs a conv with num_features = out_channels "
                                 f"(got {type(conv).__name__}, num_features {block['num_features']})")
            self.replaced_block_names.append(name)
            ref = next(self.student.parameters())
            self.added_gates[name] = GateLayer(block['num_features']).to(ref.device)

    def _student_engine(self):
        eng = super()._student_engine()
        eng.probe_names = list(self.added_gates)
        return eng

    def reset(self):
        self._remove_hooks()
        self.hint_block_names = []
        self.replaced_block_names = []
        self.added_gates = dict()

    def get_gate_importance(self):
        """{gate name: (gate * d loss/d gate)^2 as a numpy vector} for the last backward (reference :58-66)."""
        eng = self._student_engine()
        out = {}
        for name, gate in self.added_gates.items():
            g = eng.probe_grads.get(name)
            if g is None:
                raise RuntimeError(f"no gradient reached the gate behind {name}: call loss.backward() on a loss of the student logits first")
            gate.weight.grad = g.detach().to(gate.weight.device).clone()
            out[name] = ((gate.weight.detach() * gate.weight.grad) ** 2).cpu().numpy()
        return out
