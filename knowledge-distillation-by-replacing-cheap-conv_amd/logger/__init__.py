"""Logging setup and an optional TensorBoard writer (reference logger/): plain `logging`, no external config file."""
import logging
import time
from pathlib import Path


def setup_logging(save_dir, default_level=logging.INFO):
    root = logging.getLogger()
    if getattr(root, "_kdcc_configured", False):
        return
    fmt = logging.Formatter('%(asctime)s - %(name)s - %(levelname)s - %(message)s')
    sh = logging.StreamHandler()
    sh.setLevel(logging.DEBUG)
    sh.setFormatter(logging.Formatter('%(message)s'))
    root.addHandler(sh)
    try:
        fh = logging.FileHandler(str(Path(save_dir) / 'info.log'))
        fh.setLevel(logging.INFO)
        fh.setFormatter(fmt)
        root.addHandler(fh)
    except OSError:
        pass
    root.setLevel(default_level)
    root._kdcc_configured = True


class TensorboardWriter:
    """Proxy that forwards add_* calls to a SummaryWriter when one is importable and enabled; also emits the
    reference's only throughput signal, `steps_per_sec`, on every set_step (logger/visualization.py:40-48)."""

    def __init__(self, log_dir, logger, enabled):
        self.writer = None
        if enabled:
            try:
                from torch.utils.tensorboard import SummaryWriter
                self.writer = SummaryWriter(str(log_dir))
            except Exception:  # tensorboard not installed
                logger.warning("TensorBoard is configured but not importable; scalars are not written.")
        self.step, self.mode, self.timer = 0, '', time.time()

    def set_step(self, step, mode='train'):
        self.mode, self.step = mode, step
        if step == 0:
            self.timer = time.time()
        else:
            now = time.time()
            self.add_scalar('steps_per_sec', 1.0 / max(now - self.timer, 1e-9))
            self.timer = now

    def add_scalar(self, tag, value, *a, **k):
        if self.writer is not None:
            self.writer.add_scalar('{}/{}'.format(tag, self.mode), value, self.step)

    def __getattr__(self, name):
        def noop(*a, **k):
            if self.writer is not None and hasattr(self.writer, name):
                return getattr(self.writer, name)(*a, **k)
        return noop
