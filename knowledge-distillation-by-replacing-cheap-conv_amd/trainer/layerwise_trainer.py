"""`LayerwiseTrainer`: the KD training loop with the reference's constructor, epoch semantics and logged keys
(trainer/layerwise_trainer.py:18-427).  Reference behaviour that is kept on purpose: loss = hint loss only (:233-235);
the student is left in eval mode (:211-215); the optimizer is rebuilt at epoch 1 from requires_grad parameters (:175-186);
`len_epoch + 1` iterations per epoch (:277-278); the plateau scheduler steps on the running mean of `loss` (:293-296).

Re-designed for the GPU: no per-step host synchronisation (the reference does five `.item()` and two full-logit D2H
copies per step, :244-250): losses are accumulated as device scalars, the confusion matrices are built on the device and
only read at log points; with torch.distributed initialised each rank runs its own shard and gradients are averaged by
parallel.GradReducer from inside backward.
"""
import gc
import os
from functools import reduce

import torch
from torch import nn

from ..base import BaseTrainer
from ..models import forgiving_state_restore
from ..parallel import mean_scalar
from ..utils import CityscapesMetricTracker, EarlyStopTracker, MetricTracker, inf_loop
from ..utils import optim as optim_module
from ..utils.optim.lr_scheduler import MyOneCycleLR, MyReduceLROnPlateau
from ..utils.plan_schedule import PlanSchedule

_LOSS_KEYS = ('loss', 'supervised_loss', 'kd_loss', 'hint_loss', 'teacher_loss')


class LayerwiseTrainer(BaseTrainer):
    def __init__(self, model, criterions, metric_ftns, optimizer, config, train_data_loader, valid_data_loader=None,
                 lr_scheduler=None, weight_scheduler=None):
        super().__init__(model, None, metric_ftns, optimizer, config)
        self.config = config
        self.train_data_loader = train_data_loader
        self.valid_data_loader = valid_data_loader
        self.do_validation = self.valid_data_loader is not None
        self.do_validation_interval = self.config['trainer']['do_validation_interval']
        self.lr_scheduler = lr_scheduler
        self.weight_scheduler = weight_scheduler
        self.log_step = config['trainer']['log_step']
        if "len_epoch" in self.config['trainer']:
            self.train_data_loader = inf_loop(train_data_loader)   # iteration-based training
            self.len_epoch = self.config['trainer']['len_epoch']
        else:
            self.len_epoch = len(self.train_data_loader)            # epoch-based training

        names = [m.__name__ for m in self.metric_ftns]
        self.train_metrics = MetricTracker(*_LOSS_KEYS, *names, writer=self.writer)
        self.train_iou_metrics = CityscapesMetricTracker(writer=self.writer)
        self.train_teacher_iou_metrics = CityscapesMetricTracker(writer=self.writer)
        self.valid_metrics = MetricTracker(*_LOSS_KEYS, *names, writer=self.writer)
        self.valid_iou_metrics = CityscapesMetricTracker(writer=self.writer)
        self.test_metrics = MetricTracker(*_LOSS_KEYS, *names, *['teacher_' + n for n in names], writer=self.writer)
        self.test_iou_metrics = CityscapesMetricTracker(writer=self.writer)
        self.val_iou_tracker = EarlyStopTracker('best', 'max', 0.01, 'rel')

        self.criterions = nn.ModuleList(criterions).to(self.device)
        del self.criterion
        self.track_miou = bool(self.config['trainer'].get('track_train_miou', True))
        self.backprop = str(self.config['trainer'].get('backprop', 'hint'))
        if self.backprop not in ('hint', 'kd+hint'):
            raise ValueError("trainer.backprop must be 'hint' (reference behaviour) or 'kd+hint'")
        if self.backprop == 'kd+hint':
            self.model.logits_need_grad = True    # (Gated-SCNN: the shape stream must keep its intermediates, engine.py)
        self._reducer = None
        if 'resume_path' in self.config['trainer']:
            self.resume(self.config['trainer']['resume_path'])

    # ------------------------------------------------------------------ epoch preparation (host side)
    def _schedule(self, config=None):
        """config['pruning'] as a table {epoch: Stage} (utils/plan_schedule.py), read once per config object."""
        config = self.config if config is None else config
        cached = self.__dict__.get('_plan_schedule')
        if cached is None or cached[0] is not config:
            cached = (config, PlanSchedule(config['pruning']))
            self._plan_schedule = cached
        return cached[1]

    def prepare_train_epoch(self, epoch, config=None):
        """Start of an epoch (reference trainer/layerwise_trainer.py:72-145): trackers and schedulers start over, then the stage the plan
        schedules for this epoch -- if any -- is applied to the student and the optimizer is made to match the new trainable set."""
        schedule = self._schedule(config)
        self.reset_scheduler()
        stage = schedule.stage(epoch)
        self.logger.info('EPOCH: ' + str(epoch))
        if stage is None:
            self.logger.info('the plan schedules nothing for this epoch')
            return
        for line in stage.describe():
            self.logger.info(line)
        if stage.train_everything:
            for param in self.model.student.parameters():
                param.requires_grad = True
        else:
            if schedule.deprecated_kwargs:
                self.logger.warning("config['pruning'] has no 'args': taking the block arguments from 'pruner' (old checkpoint)")
            self.model.replace(stage.replace, **schedule.block_kwargs)
            self.model.register_hint_layers(stage.hints)
            self.model.unfreeze(stage.unfreeze_names)
        if epoch == 1:
            self.create_new_optimizer()           # fresh optimizer and LR scheduler: the ones passed to __init__ never saw these parameters
        else:
            self.update_optimizer(stage.unfreeze)
        self._reducer = None                      # trainable set changed: rebuild the gradient buckets lazily
        self.logger.info(self.model.dump_trainable_params())
        if not stage.train_everything:
            self.logger.info(self.model.dump_student_teacher_blocks_info())

    def update_optimizer(self, unfreeze_config):
        """One new param group per layer unfrozen after epoch 1 (entries {'name', 'epoch', ['lr']}): the optimizer's configured
        arguments, with the entry's own learning rate when it names one (reference :152-173)."""
        for entry in unfreeze_config:
            group = dict(self.config['optimizer']['args'])
            if 'lr' in entry:
                group['lr'] = entry['lr']
                self.config['optimizer']['args']['lr'] = entry['lr']   # (the reference overwrites its config dict: later groups inherit it)
            group['params'] = list(self.model.get_block(entry['name'], self.model.student).parameters())
            self.logger.debug('optimizer: + {} ({} tensors, lr {})'.format(entry['name'], len(group['params']), group.get('lr')))
            self.optimizer.add_param_group(group)

    def create_new_optimizer(self):
        """Optimizer and LR scheduler built from the config over what is trainable NOW (reference :175-188)."""
        trainable = [p for p in self.model.student.parameters() if p.requires_grad]
        self.optimizer = self.config.init_obj('optimizer', optim_module, trainable)
        self.lr_scheduler = self.config.init_obj('lr_scheduler', optim_module.lr_scheduler, self.optimizer)
        self.logger.debug('optimizer rebuilt over {} tensors'.format(len(trainable)))

    def reset_scheduler(self):
        """Everything that tracks progress within a stage starts over (reference :190-201)."""
        for tracker in (self.weight_scheduler, self.val_iou_tracker, self.train_metrics, self.valid_metrics, self.train_iou_metrics,
                        self.valid_iou_metrics, self.train_teacher_iou_metrics):
            tracker.reset()
        if isinstance(self.lr_scheduler, MyReduceLROnPlateau):
            self.lr_scheduler.reset()

    # ------------------------------------------------------------------ data-parallel plumbing
    def _attach_reducer(self):
        if self.world_size == 1 or self._reducer is not None:
            return
        from ..parallel import GradReducer, broadcast_module
        broadcast_module(self.model.student)   # replicas start from rank 0's weights (new blocks come from each rank's RNG)
        if getattr(self.model, "fused", False):
            eng = self.model._student_engine()
            self._reducer = GradReducer(eng.grad_production_order())
            eng.reducer = self._reducer
        else:
            params = [p for p in reversed(list(self.model.student.parameters())) if p.requires_grad]
            self._reducer = GradReducer(params)

    def _reduce_unfused_grads(self):
        """Non-fused students (plain module graphs): all-reduce after backward."""
        if self.world_size == 1 or getattr(self.model, "fused", False):
            return
        for p in self._reducer.params:
            self._reducer.grad_buffer(p).copy_(p.grad)
            self._reducer.grad_ready(p)
        self._reducer.finish()
        for p in self._reducer.params:
            p.grad.copy_(self._reducer.grad_buffer(p))

    # ------------------------------------------------------------------ the hot loop
    def _filter_weight(self, index, channels, device):
        """Per-filter weights for WeightedHintMSELoss.  The reference never supplies them (its trainers call the criterion
        with two arguments), so the source is build-defined: config['trainer']['hint_filter_weight'] =
        'uniform' (default) | 'rand:<seed>' (rand(C) with generator seed+index, BASELINE config 4) | path to a torch-saved
        {hint name: (C,) tensor} dict, e.g. averaged Taylor importances."""
        key = (index, channels)
        cache = self.__dict__.setdefault('_fw_cache', {})
        if key not in cache:
            spec = str(self.config['trainer'].get('hint_filter_weight', 'uniform'))
            if spec == 'uniform':
                w = torch.ones(channels)
            elif spec.startswith('rand:'):
                w = torch.rand(channels, generator=torch.Generator().manual_seed(int(spec[5:]) + index))
            else:
                table = self.__dict__.setdefault('_fw_table', None) or torch.load(spec, map_location='cpu')
                self._fw_table = table
                names = getattr(self.model, 'student_hint_names', None) or self.model.hint_block_names
                w = torch.as_tensor(table[names[index]]).float()
            cache[key] = w.to(device)
        return cache[key]

    def _hint_loss(self):
        from ..losses import WeightedHintMSELoss
        pairs = list(zip(self.model.student_hidden_outputs, self.model.teacher_hidden_outputs))
        if isinstance(self.criterions[2], WeightedHintMSELoss):
            return reduce(lambda acc, ist: acc + self.criterions[2](ist[1][0], ist[1][1],
                                                                    self._filter_weight(ist[0], ist[1][0].shape[1], ist[1][0].device)),
                          enumerate(pairs), 0)
        return reduce(lambda acc, st: acc + self.criterions[2](st[0], st[1]), pairs, 0)

    def _train_epoch(self, epoch):
        self.prepare_train_epoch(epoch)
        # the student deliberately stays in eval mode (reference :211-215); hints must be collected even after a
        # validation pass switched them off (reference latent bug)
        self.model.save_hidden = True
        self.train_iou_metrics.reset()
        self.train_teacher_iou_metrics.reset()
        self._clean_cache()
        self._attach_reducer()

        # trainer.teacher_overlap = "backward" (opt-in): the teacher's forward for batch i + 1 is launched on the side stream right
        # before batch i's loss.backward() (DepthwiseStudent.prefetch_teacher), which needs batch i + 1 on the device one step early
        lookahead = str(self.config['trainer'].get('teacher_overlap', 'none')) == 'backward' and hasattr(self.model, 'prefetch_teacher')
        for batch_idx, (data, target, next_data) in enumerate(self._device_batches(self.train_data_loader, lookahead)):
            output_st, output_tc = self.model(data)

            supervised_loss = self.criterions[0](output_st, target) / self.accumulation_steps
            kd_loss = self.criterions[1](output_st, output_tc) / self.accumulation_steps
            teacher_loss = self.criterions[0](output_tc, target)   # for comparison
            hint_loss = self._hint_loss() / self.accumulation_steps

            loss = hint_loss                                        # only use hint loss (reference :233-235)
            if self.backprop == 'kd+hint':                          # SURVEY 8(d) mode B, opt-in: trainer.backprop
                loss = kd_loss + hint_loss
            if next_data is not None and batch_idx != self.len_epoch:
                self.model.prefetch_teacher(next_data)
            loss.backward()
            self._reduce_unfused_grads()
            if batch_idx % self.accumulation_steps == 0:
                self.optimizer.step()
                self.optimizer.zero_grad()
            self.writer.set_step((epoch - 1) * self.len_epoch + batch_idx)

            acc = self.accumulation_steps
            self.train_metrics.update('loss', loss.detach() * acc)
            self.train_metrics.update('supervised_loss', supervised_loss.detach() * acc)
            self.train_metrics.update('kd_loss', kd_loss.detach() * acc)
            self.train_metrics.update('hint_loss', hint_loss.detach() * acc)
            self.train_metrics.update('teacher_loss', teacher_loss.detach())
            if self.track_miou:
                self.train_iou_metrics.update(output_st, target)
                self.train_teacher_iou_metrics.update(output_tc, target)
            for met in self.metric_ftns:
                self.train_metrics.update(met.__name__, met(output_st, target))

            if batch_idx % self.log_step == 0:
                self.train_metrics.flush()   # buffered device scalars -> TensorBoard, one host sync per log point
            if batch_idx % self.log_step == 0 and self.rank == 0:
                self.logger.info(
                    'Train Epoch: {} [{}]/[{}] Loss: {:.6f} mIoU: {:.6f} Teacher mIoU: {:.6f} Supervised Loss: {:.6f} '
                    'Knowledge Distillation loss: {:.6f} Hint Loss: {:.6f} Teacher Loss: {:.6f}'.format(
                        epoch, batch_idx, self.len_epoch, self.train_metrics.avg('loss'), self.train_iou_metrics.get_iou(),
                        self.train_teacher_iou_metrics.get_iou(), self.train_metrics.avg('supervised_loss'),
                        self.train_metrics.avg('kd_loss'), self.train_metrics.avg('hint_loss'),
                        self.train_metrics.avg('teacher_loss')))
            if batch_idx == self.len_epoch:
                break

        self.train_metrics.flush()
        log = self.train_metrics.result()
        log.update({'train_teacher_mIoU': self.train_teacher_iou_metrics.get_iou()})
        log.update({'train_student_mIoU': self.train_iou_metrics.get_iou()})
        if self.do_validation and ((epoch % self.config["trainer"]["do_validation_interval"]) == 0):
            val_log = self._valid_epoch(epoch)
            log.update(**{'val_' + k: v for k, v in val_log.items()})
            log.update(**{'val_mIoU': self.valid_iou_metrics.get_iou()})
            self.val_iou_tracker.update(self.valid_iou_metrics.get_iou())
        self._teacher_student_iou_gap = self.train_teacher_iou_metrics.get_iou() - self.train_iou_metrics.get_iou()

        if (self.lr_scheduler is not None) and (not isinstance(self.lr_scheduler, MyOneCycleLR)):
            if isinstance(self.lr_scheduler, MyReduceLROnPlateau):
                # every rank sees its own shard's loss: decide on the rank mean so the replicas cut the LR together
                self.lr_scheduler.step(mean_scalar(self.train_metrics.avg('loss')))
            else:
                self.lr_scheduler.step()
        self.weight_scheduler.step()
        return log

    def _device_batches(self, loader, lookahead):
        """(data, target, next_data) with the tensors on the device; with `lookahead` the following batch's data is transferred one
        step early and handed out as next_data (None for the last batch, and always None without lookahead)."""
        to = lambda t: t.to(self.device, non_blocking=True)
        if not lookahead:
            for data, target in loader:
                yield to(data), to(target), None
            return
        it = iter(loader)
        try:
            data, target = next(it)
        except StopIteration:
            return
        cur = (to(data), to(target))
        for data, target in it:
            nxt = (to(data), to(target))
            yield cur[0], cur[1], nxt[0]
            cur = nxt
        yield cur[0], cur[1], None

    def _valid_epoch(self, epoch):
        self._clean_cache()
        self.model.save_hidden = False
        self.valid_metrics.reset()
        self.valid_iou_metrics.reset()
        with torch.no_grad():
            for batch_idx, (data, target) in enumerate(self.valid_data_loader):
                data, target = data.to(self.device), target.to(self.device)
                output = self.model.inference(data)
                supervised_loss = self.criterions[0](output, target)
                self.writer.set_step((epoch - 1) * len(self.valid_data_loader) + batch_idx, 'valid')
                self.valid_metrics.update('supervised_loss', supervised_loss)
                self.valid_iou_metrics.update(output, target)
                for met in self.metric_ftns:
                    self.valid_metrics.update(met.__name__, met(output, target))
        self.valid_metrics.flush()   # buffered device scalars -> TensorBoard (the reference writes them every step)
        result = self.valid_metrics.result()
        result['mIoU'] = self.valid_iou_metrics.get_iou()
        return result

    def _test_epoch(self, epoch):
        """Sliding-window + flip inference of the student over the validation loader (reference :340-376): batches are
        (image names, data, target); config['test']['args'] = {scales, crop_size}; with config['submission']['save_output']
        the arg-max maps are written as label-id PNGs (needs PIL and a dataset exposing id_to_trainid)."""
        self._clean_cache()
        self.model.save_hidden = False
        self.test_metrics.reset()
        self.test_iou_metrics.reset()
        args = self.config['test']['args']
        sub = self.config['submission'] if 'submission' in self.config else {'save_output': False}
        if sub.get('save_output'):
            os.makedirs(sub['path_output'], exist_ok=True)
        with torch.no_grad():
            for batch_idx, (img_name, data, target) in enumerate(self.valid_data_loader):
                data, target = data.to(self.device), target.to(self.device)
                output = self.model.inference_test(data, args)
                if sub.get('save_output'):
                    self.save_for_submission(output, img_name[0])
                self.writer.set_step((epoch - 1) * len(self.valid_data_loader) + batch_idx, 'test')
                self.test_metrics.update('supervised_loss', self.criterions[0](output, target))
                self.test_iou_metrics.update(output, target)
                for met in self.metric_ftns:
                    self.test_metrics.update(met.__name__, met(output, target))
        self.test_metrics.flush()
        result = self.test_metrics.result()
        result['mIoU'] = self.test_iou_metrics.get_iou()
        return result

    def save_for_submission(self, output, image_name):
        """arg-max -> dataset label ids (trainId -> id through the dataset's id_to_trainid, reference :378-397) -> PNG."""
        from PIL import Image
        sub = self.config['submission']
        pred = torch.argmax(output, dim=1)
        mapping = getattr(getattr(self.valid_data_loader, 'dataset', None), 'id_to_trainid', None)
        out = torch.zeros_like(pred)
        if mapping is not None:
            for k, v in mapping.items():
                out[pred == v] = k
        else:
            out = pred
        arr = out[0].to(torch.uint8).cpu().numpy()
        Image.fromarray(arr).save(os.path.join(sub['path_output'], '{}.{}'.format(image_name, sub['ext'])))

    def _clean_cache(self):
        self.model.student_hidden_outputs, self.model.teacher_hidden_outputs = list(), list()
        gc.collect()

    def resume(self, checkpoint_path):
        """Rebuild the replaced-block topology by replaying prepare_train_epoch for every saved epoch, then load weights."""
        self.logger.info("Loading checkpoint: {} ...".format(checkpoint_path))
        checkpoint = torch.load(checkpoint_path, map_location=torch.device('cpu'), weights_only=False)
        self.start_epoch = checkpoint['epoch'] + 1
        self.mnt_best = checkpoint['monitor_best']
        config = checkpoint['config']
        for i in range(1, checkpoint['epoch'] + 1):
            self.prepare_train_epoch(i, config)
        forgiving_state_restore(self.model, checkpoint['state_dict'])
        self.logger.info("Loaded model's state dict")
        if checkpoint['config']['optimizer']['type'] != self.config['optimizer']['type']:
            self.logger.warning("Warning: Optimizer type given in config file is different from that of checkpoint. "
                                "Optimizer parameters not being resumed.")
        elif self.optimizer is not None and checkpoint.get('optimizer') is not None:
            self.optimizer.load_state_dict(checkpoint['optimizer'])
        self.logger.info("Checkpoint loaded. Resume training from epoch {}".format(self.start_epoch))
