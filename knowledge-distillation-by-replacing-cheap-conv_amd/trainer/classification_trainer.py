"""`ClassificationTrainer` (reference trainer/classification_trainer.py:7-116): same loop, but the student is put in
train mode (batch-statistics BN), the back-propagated loss is the KD loss, and the optimizer steps on
(batch_idx + 1) % accumulation_steps."""
from functools import reduce

import torch

from ..parallel import mean_scalar
from ..utils import MetricTracker
from ..utils.optim.lr_scheduler import MyOneCycleLR, MyReduceLROnPlateau
from .layerwise_trainer import _LOSS_KEYS, LayerwiseTrainer


class ClassificationTrainer(LayerwiseTrainer):
    def __init__(self, model, criterions, metric_ftns, optimizer, config, train_data_loader, valid_data_loader=None,
                 lr_scheduler=None, weight_scheduler=None, test_data_loader=None):
        super().__init__(model, criterions, metric_ftns, optimizer, config, train_data_loader, valid_data_loader,
                         lr_scheduler, weight_scheduler)
        names = [m.__name__ for m in self.metric_ftns]
        self.train_teacher_metrics = MetricTracker(*names, writer=self.writer)
        self.valid_metrics = MetricTracker(*_LOSS_KEYS, *names, *['teacher_' + n for n in names], writer=self.writer)
        self.test_data_loader = test_data_loader

    def _train_epoch(self, epoch):
        self.prepare_train_epoch(epoch)
        self.model.train()
        self._clean_cache()
        self._attach_reducer()
        for batch_idx, (data, target) in enumerate(self.train_data_loader):
            data, target = data.to(self.device), target.to(self.device)
            output_st, output_tc = self.model(data)
            supervised_loss = self.criterions[0](output_st, target) / self.accumulation_steps
            kd_loss = self.criterions[1](output_st, output_tc) / self.accumulation_steps
            pairs = zip(self.model.student_hidden_outputs, self.model.teacher_hidden_outputs)
            hint_loss = reduce(lambda acc, st: acc + self.criterions[2](st[0], st[1]), pairs,
                               torch.tensor(0., device=self.device)) / self.accumulation_steps
            teacher_loss = self.criterions[0](output_tc, target)

            loss = kd_loss
            loss.backward()
            self._reduce_unfused_grads()
            if (batch_idx + 1) % self.accumulation_steps == 0:
                self.optimizer.step()
                self.optimizer.zero_grad()
            self.writer.set_step((epoch - 1) * self.len_epoch + batch_idx)
            acc = self.accumulation_steps
            self.train_metrics.update('loss', loss.detach() * acc)
            self.train_metrics.update('supervised_loss', supervised_loss.detach() * acc)
            self.train_metrics.update('kd_loss', kd_loss.detach() * acc)
            self.train_metrics.update('hint_loss', hint_loss.detach() * acc)
            self.train_metrics.update('teacher_loss', teacher_loss.detach())
            for met in self.metric_ftns:
                self.train_metrics.update(met.__name__, met(output_st, target), data.shape[0])
                self.train_teacher_metrics.update(met.__name__, met(output_tc, target), data.shape[0])
            if batch_idx % self.log_step == 0:
                self.train_metrics.flush()
            if batch_idx % self.log_step == 0 and self.rank == 0:
                self.logger.info('Train Epoch: {} [{}]/[{}] Loss: {:.6f} Supervised Loss: {:.6f} Knowledge Distillation '
                                 'loss: {:.6f} Hint Loss: {:.6f} Teacher Loss: {:.6f}'.format(
                                     epoch, batch_idx, self.len_epoch, self.train_metrics.avg('loss'),
                                     self.train_metrics.avg('supervised_loss'), self.train_metrics.avg('kd_loss'),
                                     self.train_metrics.avg('hint_loss'), self.train_metrics.avg('teacher_loss')))
            if batch_idx == self.len_epoch:
                break
        self.train_metrics.flush()
        log = self.train_metrics.result()
        if self.do_validation and ((epoch % self.config["trainer"]["do_validation_interval"]) == 0):
            val_log = self._valid_epoch(epoch)
            log.update(**{'val_' + k: v for k, v in val_log.items()})
        if (self.lr_scheduler is not None) and (not isinstance(self.lr_scheduler, MyOneCycleLR)):
            if isinstance(self.lr_scheduler, MyReduceLROnPlateau):
                self.lr_scheduler.step(mean_scalar(self.train_metrics.avg('loss')))
            else:
                self.lr_scheduler.step()
        self.weight_scheduler.step()
        return log

    def _valid_epoch(self, epoch):
        self._clean_cache()
        self.model.eval()
        self.valid_metrics.reset()
        with torch.no_grad():
            for batch_idx, (data, target) in enumerate(self.valid_data_loader):
                data, target = data.to(self.device), target.to(self.device)
                output_st, output_tc = self.model(data)
                self.valid_metrics.update('supervised_loss', self.criterions[0](output_st, target))
                self.valid_metrics.update('kd_loss', self.criterions[1](output_st, output_tc))
                self.valid_metrics.update('teacher_loss', self.criterions[0](output_tc, target))
                for met in self.metric_ftns:
                    self.valid_metrics.update(met.__name__, met(output_st, target), data.shape[0])
                    self.valid_metrics.update('teacher_' + met.__name__, met(output_tc, target), data.shape[0])
        self.valid_metrics.flush()   # buffered device scalars -> TensorBoard
        return self.valid_metrics.result()
