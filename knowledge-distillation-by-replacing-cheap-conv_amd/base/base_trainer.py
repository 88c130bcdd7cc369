"""`BaseTrainer`: epoch loop, best-metric monitoring / early stop, checkpoint save and resume (reference
base/base_trainer.py:7-212).  Device placement is one process per GPU: the trainer uses cuda:LOCAL_RANK and, when
torch.distributed is initialised, exchanges gradients through parallel.GradReducer instead of nn.DataParallel
(which cannot work with these trainers)."""
import os
from abc import abstractmethod

import torch
import torch.distributed as dist
from numpy import inf

from ..logger import TensorboardWriter


class BaseTrainer:
    def __init__(self, model, criterion, metric_ftns, optimizer, config):
        self.config = config
        self.logger = config.get_logger('trainer', config['trainer']['verbosity'])
        self.device, self.world_size, self.rank = self._prepare_device(config['n_gpu'])
        self.model = model.to(self.device)
        self.criterion = criterion
        self.metric_ftns = metric_ftns
        self.optimizer = optimizer

        cfg_trainer = config['trainer']
        self.accumulation_steps = cfg_trainer['accumulation_steps']
        self.epochs = cfg_trainer['epochs']
        self.save_period = cfg_trainer['save_period']
        self.monitor = cfg_trainer.get('monitor', 'off')
        if self.monitor == 'off':
            self.mnt_mode = 'off'
            self.mnt_best = 0
        else:
            self.mnt_mode, self.mnt_metric = self.monitor.split()
            assert self.mnt_mode in ['min', 'max']
            self.mnt_best = inf if self.mnt_mode == 'min' else -inf
            self.early_stop = cfg_trainer.get('early_stop', inf)
        self.start_epoch = 1
        self.checkpoint_dir = config.save_dir
        self.writer = TensorboardWriter(config.log_dir, self.logger, cfg_trainer['tensorboard'] and self.rank == 0)
        if config.resume is not None:
            self._resume_checkpoint(config.resume)

    @abstractmethod
    def _train_epoch(self, epoch):
        raise NotImplementedError

    @abstractmethod
    def _valid_epoch(self, epoch):
        raise NotImplementedError

    @abstractmethod
    def _test_epoch(self, epoch):
        raise NotImplementedError

    def train(self):
        not_improved_count = 0
        for epoch in range(self.start_epoch, self.epochs + 1):
            result = self._train_epoch(epoch)
            log = {'epoch': epoch}
            log.update(result)
            for key, value in log.items():
                self.logger.info('    {:15s}: {}'.format(str(key), value))
            best = False
            if self.mnt_mode != 'off':
                try:
                    improved = (self.mnt_mode == 'min' and log[self.mnt_metric] <= self.mnt_best) or \
                               (self.mnt_mode == 'max' and log[self.mnt_metric] >= self.mnt_best)
                except KeyError:
                    self.logger.warning("Warning: Metric '{}' is not found. Model performance monitoring is disabled."
                                        .format(self.mnt_metric))
                    self.mnt_mode = 'off'
                    improved = False
                if improved:
                    self.mnt_best = log[self.mnt_metric]
                    not_improved_count = 0
                    best = True
                else:
                    not_improved_count += 1
                if not_improved_count > self.early_stop:
                    self.logger.info("Validation performance didn't improve for {} epochs. Training stops."
                                     .format(self.early_stop))
                    break
            if epoch % self.save_period == 0:
                self._save_checkpoint(epoch, save_best=best)

    def eval(self):
        for key, value in self._valid_epoch(1).items():
            self.logger.info('    {:15s}: {}'.format(str(key), value))

    def test(self):
        for key, value in self._test_epoch(1).items():
            self.logger.info('    {:15s}: {}'.format(str(key), value))

    def _prepare_device(self, n_gpu_use):
        """One process drives one device.  n_gpu == 0 (or no GPU) -> CPU, like the reference; otherwise cuda:LOCAL_RANK."""
        world = dist.get_world_size() if dist.is_initialized() else 1
        rank = dist.get_rank() if dist.is_initialized() else 0
        if n_gpu_use > 0 and not torch.cuda.is_available():
            self.logger.warning("Warning: There's no GPU available on this machine, training will be performed on CPU.")
            n_gpu_use = 0
        if n_gpu_use == 0:
            return torch.device('cpu'), world, rank
        local = int(os.environ.get("LOCAL_RANK", "0"))
        torch.cuda.set_device(local)
        return torch.device('cuda', local), world, rank

    def _save_checkpoint(self, epoch, save_best=False):
        if self.rank != 0:
            return
        state = {'arch': type(self.model).__name__, 'epoch': epoch, 'state_dict': self.model.state_dict(),
                 'optimizer': self.optimizer.state_dict(), 'monitor_best': self.mnt_best, 'config': self.config}
        filename = str(self.checkpoint_dir / 'checkpoint-epoch{}.pth'.format(epoch))
        torch.save(state, filename)
        self.logger.info("Saving checkpoint: {} ...".format(filename))
        if save_best:
            torch.save(state, str(self.checkpoint_dir / 'model_best.pth'))
            self.logger.info("Saving current best: model_best.pth ...")

    def _resume_checkpoint(self, resume_path):
        resume_path = str(resume_path)
        self.logger.info("Loading checkpoint: {} ...".format(resume_path))
        checkpoint = torch.load(resume_path, map_location='cpu', weights_only=False)
        self.start_epoch = checkpoint['epoch'] + 1
        self.mnt_best = checkpoint['monitor_best']
        self.model.load_state_dict(checkpoint['state_dict'])
        if checkpoint['config']['optimizer']['type'] != self.config['optimizer']['type']:
            self.logger.warning("Warning: Optimizer type given in config file is different from that of checkpoint. "
                                "Optimizer parameters not being resumed.")
        else:
            self.optimizer.load_state_dict(checkpoint['optimizer'])
        self.logger.info("Checkpoint loaded. Resume training from epoch {}".format(self.start_epoch))
