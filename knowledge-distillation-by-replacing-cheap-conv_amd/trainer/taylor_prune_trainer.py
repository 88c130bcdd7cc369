"""`TaylorPruneTrainer`: ranks the filters behind gated blocks by their first-order Taylor importance while the student runs
on the supervised loss (reference trainer/taylor_prune_trainer.py:18-288).  Per step: loss = CrossEntropyLoss2d(student
logits, target); loss.backward(); importance = (gate * gate.grad)^2 goes to ImportanceFilterTracker; the optimizer -- built at
epoch 1 from the student parameters that require grad, i.e. the gates (:152-162) -- steps and zeroes the gradients every
`accumulation_steps` batches (:212-215; with the shipped accumulation_steps = 100000 that is batch 0 of each epoch only, so
gate.grad keeps accumulating in between, exactly as autograd does there).  Every `importance_log_interval` steps the running
table is dumped as `importance_filter_ep{E}_batch_idx{B}.pth` -- what `trainer.hint_filter_weight` of LayerwiseTrainer
consumes for WeightedHintMSELoss (BASELINE config 4's principled weights).

Same constructor and config keys as the reference; cfg/taylor_importance_track.json runs as shipped: `pruning` needs only
`args` and `pruning_plan` (`unfreeze` is optional here -- the reference indexes it unconditionally and raises KeyError at the
second plan epoch; gates added at a later plan epoch join the optimizer as a new param group, so every gate is stepped and
zeroed on the same schedule), validation / best-model monitoring / the plateau scheduler follow LayerwiseTrainer's epoch tail, and with
several ranks the gate gradients are averaged like every other gradient (the engine's GradReducer), so every rank accumulates
the same importances."""
import os

import torch

from ..parallel import mean_scalar
from ..utils import ImportanceFilterTracker
from ..utils.optim.lr_scheduler import MyOneCycleLR, MyReduceLROnPlateau
from .layerwise_trainer import LayerwiseTrainer
from ..utils.plan_schedule import PlanSchedule


class TaylorPruneTrainer(LayerwiseTrainer):
    def __init__(self, model, criterions, metric_ftns, optimizer, config, train_data_loader, valid_data_loader=None,
                 lr_scheduler=None, weight_scheduler=None):
        super().__init__(model, criterions, metric_ftns, optimizer, config, train_data_loader, valid_data_loader, lr_scheduler,
                         weight_scheduler)
        self.importance_log_interval = self.config['trainer']['importance_log_interval']
        self.model.logits_need_grad = True        # the supervised loss is back-propagated through the logits
        self.importance_tracker = ImportanceFilterTracker(writer=self.writer)

    def create_new_optimizer(self):
        if any(p.requires_grad for p in self.model.student.parameters()):
            super().create_new_optimizer()
        else:
            self.optimizer = None   # nothing trainable (a plan without gates): importance tracking only

    def _schedule(self, config=None):
        config = self.config if config is None else config
        cached = self.__dict__.get('_plan_schedule')
        if cached is None or cached[0] is not config:
            cached = (config, PlanSchedule(config['pruning'], which=('pruning_plan',)))   # only the plan opens a stage here (reference :75-131)
            self._plan_schedule = cached
        return cached[1]

    def prepare_train_epoch(self, epoch, config=None):
        """Gate the layers the plan schedules for `epoch`; trackers start over afterwards."""
        schedule = self._schedule(config)
        stage = schedule.stage(epoch)
        self.logger.info('EPOCH: ' + str(epoch))
        if stage is None or stage.train_everything:
            self.logger.info('the plan schedules nothing for this epoch')
            return
        self.logger.info('gate ' + str(stage.replace))
        if schedule.deprecated_kwargs:
            self.logger.warning("config['pruning'] has no 'args': taking the gate arguments from 'pruner' (old checkpoint)")
        self.model.replace(stage.replace, **schedule.block_kwargs)
        self.importance_tracker.update_importance_list(self.model.added_gates)
        if epoch == 1 or self.optimizer is None:
            self.create_new_optimizer()
        else:
            # The gates added at this plan epoch join the optimizer as a new group: they are stepped and -- what the importances
            # depend on -- ZEROED with the earlier ones (the optimizer was built at epoch 1 from the gates that existed then;
            # a gate outside it would keep accumulating .grad over every batch of every later epoch).  The reference defines
            # nothing here: it raises KeyError at its second plan epoch (:101).
            known = {id(p) for g in self.optimizer.param_groups for p in g['params']}
            fresh = [p for p in self.model.student.parameters() if p.requires_grad and id(p) not in known]
            if fresh:
                self.logger.debug('optimizer: + {} new gate tensors'.format(len(fresh)))
                self.optimizer.add_param_group({'params': fresh, **self.config['optimizer']['args']})
            fresh_ids = {id(f) for f in fresh}
            self.update_optimizer([e for e in stage.unfreeze
                                   if not any(id(q) in fresh_ids for q in self.model.get_block(e['name'], self.model.student).parameters())])
        self._reducer = None              # trainable set changed: rebuild the gradient buckets lazily
        self.logger.info(self.model.dump_trainable_params())
        self.logger.info(self.model.dump_student_teacher_blocks_info())
        self.reset_scheduler()

    def _train_epoch(self, epoch):
        self.prepare_train_epoch(epoch)
        self.model.save_hidden = True
        self.train_metrics.reset()
        self.train_iou_metrics.reset()
        self.train_teacher_iou_metrics.reset()
        self._clean_cache()
        self._attach_reducer()
        for batch_idx, (data, target) in enumerate(self.train_data_loader):
            data, target = data.to(self.device, non_blocking=True), target.to(self.device, non_blocking=True)
            output_st, output_tc = self.model(data)
            supervised_loss = self.criterions[0](output_st, target)      # not divided: keeps the gradient's scale (:201)
            teacher_loss = self.criterions[0](output_tc, target)
            loss = supervised_loss                                       # only the supervised loss (:204-206)
            loss.backward()
            self._reduce_unfused_grads()
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
            for met in self.metric_ftns:
                self.train_metrics.update(met.__name__, met(output_st, target))
            if batch_idx % self.log_step == 0:
                self.train_metrics.flush()
                if self.rank == 0:
                    self.logger.info(
                        'Train Epoch: {} [{}]/[{}] Loss: {:.6f} mIoU: {:.6f} Teacher mIoU: {:.6f} Supervised Loss: {:.6f} '
                        'Teacher Loss: {:.6f}'.format(epoch, batch_idx, self.len_epoch, self.train_metrics.avg('loss'),
                                                      self.train_iou_metrics.get_iou(), self.train_teacher_iou_metrics.get_iou(),
                                                      self.train_metrics.avg('supervised_loss'), self.train_metrics.avg('teacher_loss')))
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
        if self.do_validation and ((epoch % self.config["trainer"]["do_validation_interval"]) == 0):
            val_log = self._valid_epoch(epoch)
            log.update(**{'val_' + k: v for k, v in val_log.items()})
            log.update(**{'val_mIoU': self.valid_iou_metrics.get_iou()})
            self.val_iou_tracker.update(self.valid_iou_metrics.get_iou())
        self._teacher_student_iou_gap = self.train_teacher_iou_metrics.get_iou() - self.train_iou_metrics.get_iou()
        if (self.lr_scheduler is not None) and (not isinstance(self.lr_scheduler, MyOneCycleLR)):
            if isinstance(self.lr_scheduler, MyReduceLROnPlateau):
                self.lr_scheduler.step(mean_scalar(self.train_metrics.avg('loss')))
            else:
                self.lr_scheduler.step()
        self.weight_scheduler.step()
        return log
