"""`TaylorPruneTrainer`: tracks the Taylor importance of the filters behind the gated convs while running the student on
the supervised loss (reference trainer/taylor_prune_trainer.py:181-288): per step loss = CrossEntropyLoss2d(student logits,
target), loss.backward(), importance = (gate * d loss/d gate)^2 accumulated by ImportanceFilterTracker and dumped as
`importance_filter_ep{E}_batch_idx{B}.pth` every `trainer.importance_log_interval` steps -- the table
`trainer.hint_filter_weight` of LayerwiseTrainer consumes for WeightedHintMSELoss (BASELINE config 4's principled weights).

Same constructor as LayerwiseTrainer; `pruning.pruning_plan` entries carry `num_features`.  The gates are probes, not
parameters (models/students/taylor_prune_student.py), so with a frozen student there is nothing for the optimizer to
update and no optimizer is built; if the plan unfreezes layers they train on the supervised loss as in the reference."""
import os

import torch

from ..utils import ImportanceFilterTracker
from .layerwise_trainer import LayerwiseTrainer


class TaylorPruneTrainer(LayerwiseTrainer):
    def __init__(self, model, criterions, metric_ftns, optimizer, config, train_data_loader, valid_data_loader=None,
                 lr_scheduler=None, weight_scheduler=None):
        super().__init__(model, criterions, metric_ftns, optimizer, config, train_data_loader, valid_data_loader, lr_scheduler,
                         weight_scheduler)
        self.importance_log_interval = self.config['trainer']['importance_log_interval']
        self.importance_tracker = ImportanceFilterTracker(writer=self.writer)

    def create_new_optimizer(self):
        if any(p.requires_grad for p in self.model.student.parameters()):
            super().create_new_optimizer()
        else:
            self.optimizer = None   # importance tracking on a frozen student

    def prepare_train_epoch(self, epoch, config=None):
        super().prepare_train_epoch(epoch, config)
        self.importance_tracker.update_importance_list(self.model.added_gates)

    def _train_epoch(self, epoch):
        self.prepare_train_epoch(epoch)
        self.model.save_hidden = True
        self.train_metrics.reset()
        self.train_iou_metrics.reset()
        self.train_teacher_iou_metrics.reset()
        self._clean_cache()
        for batch_idx, (data, target) in enumerate(self.train_data_loader):
            data, target = data.to(self.device), target.to(self.device)
            output_st, output_tc = self.model(data)
            supervised_loss = self.criterions[0](output_st, target)      # not divided: keeps the gradient's scale
            teacher_loss = self.criterions[0](output_tc, target)
            loss = supervised_loss                                       # only the supervised loss
            loss.backward()
            self.importance_tracker.update(self.model.get_gate_importance())
            if self.optimizer is not None and batch_idx % self.accumulation_steps == 0:
                self.optimizer.step()
                self.optimizer.zero_grad()
            self.writer.set_step((epoch - 1) * self.len_epoch + batch_idx)
            self.train_metrics.update('loss', loss.detach() * self.accumulation_steps)
            self.train_metrics.update('supervised_loss', supervised_loss.detach() * self.accumulation_steps)
            self.train_metrics.update('teacher_loss', teacher_loss.detach())
            if self.track_miou:
                self.train_iou_metrics.update(output_st, target)
                self.train_teacher_iou_metrics.update(output_tc, target)
            if batch_idx % self.importance_log_interval == 0 and self.rank == 0:
                table = self.importance_tracker.average()
                path = os.path.join(str(self.checkpoint_dir), 'importance_filter_ep{}_batch_idx{}.pth'.format(epoch, batch_idx))
                torch.save(table, path)
                self.logger.info('Importance of filters in layers -> {}'.format(path))
            if batch_idx == self.len_epoch:
                break
        self.train_metrics.flush()
        log = self.train_metrics.result()
        log.update({'train_teacher_mIoU': self.train_teacher_iou_metrics.get_iou(),
                    'train_student_mIoU': self.train_iou_metrics.get_iou()})
        self.weight_scheduler.step()
        return log
