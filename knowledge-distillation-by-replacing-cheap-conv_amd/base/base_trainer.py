"""`BaseTrainer`: the epoch driver underneath the KD trainers.

Interface kept from the reference (base/base_trainer.py:7-212) because train.py / eval.py / test.py and the
trainers' subclasses rely on it: constructor `(model, criterion, metric_ftns, optimizer, config)`, `train()`,
`eval()`, `test()`, the `_train_epoch / _valid_epoch / _test_epoch` hooks, the `monitor: "<min|max> <metric>"` /
`early_stop` / `save_period` config keys, and the checkpoint dictionary
`{'arch','epoch','state_dict','optimizer','monitor_best','config'}` written as `checkpoint-epoch{N}.pth` /
`model_best.pth` under `config.save_dir`.

What is different: one process drives one device (cuda:LOCAL_RANK); with WORLD_SIZE > 1 the process group is created
here (RCCL on the GPU box, gloo on CPU), the decisions every rank must take identically -- "did the monitored metric
improve", "stop now" -- are taken on rank-averaged values, and only rank 0 touches the filesystem.  The reference's
nn.DataParallel wrapping cannot work with these trainers and is not reproduced.
"""
import os
from abc import abstractmethod

import torch

from .. import parallel
from ..logger import TensorboardWriter


class BestTracker:
    """`monitor` bookkeeping: is the new value at least as good as the best one seen, and for how many epochs in a
    row it was not."""

    def __init__(self, spec, patience=float('inf')):
        self.enabled = spec != 'off'
        self.mode, self.metric = (spec.split() if self.enabled else ('off', None))
        if self.enabled and self.mode not in ('min', 'max'):
            raise ValueError(f"monitor mode must be 'min' or 'max', got {self.mode!r}")
        self.best = 0 if not self.enabled else (float('inf') if self.mode == 'min' else -float('inf'))
        self.patience = patience
        self.stale = 0

    def observe(self, value):
        """Returns True when `value` ties or beats the best so far."""
        better = value <= self.best if self.mode == 'min' else value >= self.best
        if better:
            self.best, self.stale = value, 0
        else:
            self.stale += 1
        return better

    @property
    def exhausted(self):
        return self.stale > self.patience


class BaseTrainer:
    def __init__(self, model, criterion, metric_ftns, optimizer, config):
        self.config = config
        cfg = config['trainer']
        self.logger = config.get_logger('trainer', cfg['verbosity'])
        self.rank, self.local_rank, self.world_size = parallel.init_distributed(use_cuda=config['n_gpu'] > 0)
        self.device = self._pick_device(config['n_gpu'])
        self.model = model.to(self.device)
        self.criterion, self.metric_ftns, self.optimizer = criterion, metric_ftns, optimizer

        self.epochs = cfg['epochs']
        self.save_period = cfg['save_period']
        self.accumulation_steps = cfg['accumulation_steps']
        self.monitor = cfg.get('monitor', 'off')
        self._best = BestTracker(self.monitor, cfg.get('early_stop', float('inf')))
        self.start_epoch = 1
        self.checkpoint_dir = config.save_dir
        self.writer = TensorboardWriter(config.log_dir, self.logger, bool(cfg['tensorboard']) and self.rank == 0)
        if config.resume is not None:
            self._resume_checkpoint(config.resume)

    # the reference exposes these as plain attributes; subclasses and checkpoints read / write them
    mnt_mode = property(lambda self: self._best.mode)
    mnt_metric = property(lambda self: self._best.metric)
    early_stop = property(lambda self: self._best.patience)

    @property
    def mnt_best(self):
        return self._best.best

    @mnt_best.setter
    def mnt_best(self, value):
        self._best.best = value

    # ------------------------------------------------------------------ hooks
    @abstractmethod
    def _train_epoch(self, epoch):
        raise NotImplementedError

    @abstractmethod
    def _valid_epoch(self, epoch):
        raise NotImplementedError

    @abstractmethod
    def _test_epoch(self, epoch):
        raise NotImplementedError

    # ------------------------------------------------------------------ drivers
    def _report(self, log):
        if self.rank == 0:
            for key, value in log.items():
                self.logger.info('    {:15s}: {}'.format(str(key), value))

    def train(self):
        for epoch in range(self.start_epoch, self.epochs + 1):
            log = {'epoch': epoch, **self._train_epoch(epoch)}
            self._report(log)
            is_best = False
            if self._best.enabled:
                if self._best.metric not in log:
                    self.logger.warning("Warning: Metric '{}' is not found. Model performance monitoring is disabled."
                                        .format(self._best.metric))
                    self._best.enabled, self._best.mode = False, 'off'
                else:
                    # every rank must reach the same verdict, or the ranks leave the loop at different epochs and the
                    # remaining ones hang in the gradient all-reduce: decide on the rank mean
                    is_best = self._best.observe(parallel.mean_scalar(log[self._best.metric]))
                    if self._best.exhausted:
                        self.logger.info("Validation performance didn't improve for {} epochs. Training stops."
                                         .format(self._best.patience))
                        break
            if epoch % self.save_period == 0:
                self._save_checkpoint(epoch, save_best=is_best)

    def eval(self):
        self._report(self._valid_epoch(1))

    def test(self):
        self._report(self._test_epoch(1))

    # ------------------------------------------------------------------ device / checkpoints
    def _pick_device(self, n_gpu):
        if n_gpu > 0 and not torch.cuda.is_available():
            self.logger.warning("Warning: There's no GPU available on this machine, training will be performed on CPU.")
            n_gpu = 0
        if n_gpu == 0:
            # host plumbing run (the reference's CPU configs): the CIFAR modules may take torch's own ops -- explicit, logged,
            # never the measured path (nn_hip.py)
            from .. import nn_hip
            nn_hip.allow_host_tensors(True)
            self.logger.warning("n_gpu = 0: host plumbing mode, stock torch CPU ops (the HIP kernels are not used)")
            return torch.device('cpu')
        torch.cuda.set_device(self.local_rank)
        return torch.device('cuda', self.local_rank)

    def _checkpoint_state(self, epoch):
        return {'arch': type(self.model).__name__, 'epoch': epoch, 'state_dict': self.model.state_dict(),
                # (a trainer without trainable parameters -- TaylorPruneTrainer over a frozen student -- has no optimizer)
                'optimizer': self.optimizer.state_dict() if self.optimizer is not None else None,
                'monitor_best': self.mnt_best, 'config': self.config}

    def _save_checkpoint(self, epoch, save_best=False):
        if self.rank != 0:   # replicas are identical; one writer
            return
        state = self._checkpoint_state(epoch)
        targets = [self.checkpoint_dir / 'checkpoint-epoch{}.pth'.format(epoch)]
        if save_best:
            targets.append(self.checkpoint_dir / 'model_best.pth')
        for path in targets:
            torch.save(state, str(path))
            self.logger.info("Saving checkpoint: {} ...".format(path))

    def _optimizer_matches(self, checkpoint):
        same = checkpoint['config']['optimizer']['type'] == self.config['optimizer']['type']
        if not same:
            self.logger.warning("Warning: Optimizer type given in config file is different from that of checkpoint. "
                                "Optimizer parameters not being resumed.")
        return same

    def _resume_checkpoint(self, resume_path):
        """`-r checkpoint.pth`: strict state-dict load (the topology must already match; after layer replacement use
        `trainer.resume_path`, LayerwiseTrainer.resume)."""
        self.logger.info("Loading checkpoint: {} ...".format(resume_path))
        checkpoint = torch.load(str(resume_path), map_location='cpu', weights_only=False)
        self.start_epoch = checkpoint['epoch'] + 1
        self.mnt_best = checkpoint['monitor_best']
        self.model.load_state_dict(checkpoint['state_dict'])
        if self.optimizer is not None and checkpoint.get('optimizer') is not None and self._optimizer_matches(checkpoint):
            self.optimizer.load_state_dict(checkpoint['optimizer'])
        self.logger.info("Checkpoint loaded. Resume training from epoch {}".format(self.start_epoch))
