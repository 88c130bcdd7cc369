"""The `pruning` section of a config as a table: epoch -> what changes in the student at the start of that epoch.

The reference walks the three lists of its `pruning` section (`pruning_plan`, `hint`, `unfreeze`) again at every epoch start
(trainer/layerwise_trainer.py:72-145, trainer/taylor_prune_trainer.py:75-131).  Here the section is read ONCE into
`PlanSchedule.stages`, {epoch: Stage}; a trainer asks `stage(epoch)` and applies what it gets.  Same JSON schema, same semantics:
an entry {'name', 'epoch', ['args'], ['lr']} takes effect at the start of its epoch; an empty section means "train a student of the
teacher's own architecture" (every parameter trainable, one stage at epoch 1); the replacement block's constructor arguments come
from `pruning.args` (older checkpoints: `pruning.pruner`)."""
from collections import OrderedDict


class Stage:
    """What one epoch start changes: layers to replace (plan entries), hint layers to register (names), layers to unfreeze (entries,
    which may carry their own 'lr')."""
    __slots__ = ("epoch", "replace", "hints", "unfreeze", "train_everything")

    def __init__(self, epoch, train_everything=False):
        self.epoch, self.replace, self.hints, self.unfreeze, self.train_everything = epoch, [], [], [], train_everything

    @property
    def unfreeze_names(self):
        return [e['name'] for e in self.unfreeze]

    def describe(self):
        if self.train_everything:
            return ['identical architecture: every student parameter trains']
        return ['replace ' + str(self.replace), 'hints ' + str(self.hints), 'unfreeze ' + str(self.unfreeze_names)]


class PlanSchedule:
    def __init__(self, pruning, which=('pruning_plan', 'hint', 'unfreeze')):
        """`which`: the lists that can open a stage (LayerwiseTrainer: all three; TaylorPruneTrainer: the plan alone)."""
        self.block_kwargs, self.deprecated_kwargs = self._kwargs(pruning)
        plan, hint, unfreeze = (list(pruning.get(k, [])) for k in ('pruning_plan', 'hint', 'unfreeze'))
        self.stages = OrderedDict()
        self.identical_architecture = not (plan or hint or unfreeze)
        if self.identical_architecture:
            self.stages[1] = Stage(1, train_everything=True)
            return
        opening = {e['epoch'] for k, lst in (('pruning_plan', plan), ('hint', hint), ('unfreeze', unfreeze)) if k in which for e in lst}
        for ep in sorted(opening):
            st = Stage(ep)
            st.replace = [e for e in plan if e['epoch'] == ep]
            st.hints = [e['name'] for e in hint if e['epoch'] == ep]
            st.unfreeze = [e for e in unfreeze if e['epoch'] == ep]
            self.stages[ep] = st

    @staticmethod
    def _kwargs(pruning):
        if 'args' in pruning:
            return pruning['args'], False
        return pruning.get('pruner', {}), True

    def stage(self, epoch):
        """The Stage that starts at `epoch`, or None when nothing changes."""
        return self.stages.get(epoch)

    @property
    def epochs(self):
        return list(self.stages)
