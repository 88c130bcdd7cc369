"""`ConfigParser`: one JSON experiment description -> the objects of a run.

The JSON schema is the drop-in surface and is unchanged from the reference (cfg/**/*.json: `teacher`,
`*_data_loader`, `optimizer`, `supervised_loss`, `kd_loss`, `hint_loss`, `lr_scheduler`, `trainer`,
`pruning{args,pruning_plan,hint,unfreeze}`, `weight_scheduler`, `metrics`, `n_gpu`, `name`), as is the small interface
train.py and the trainers use (reference parse_config.py:11-175): `ConfigParser(config, resume, modification, run_id)`,
`ConfigParser.from_args(argparse, options)`, `config[name]`, `init_obj` / `init_ftn` / `restore_snapshot`,
`get_logger`, `save_dir`, `log_dir`, `resume`.

Run directories are `<save_dir>/{models,log}/<name>/<run_id>/`.  With one process per GPU every rank must agree on
`run_id`: rank 0 picks the timestamp and broadcasts it, creates the directories and writes `config.json`; the others
only make sure the directories exist.
"""
import logging
import os
from datetime import datetime
from functools import partial
from pathlib import Path

from . import parallel
from .logger import setup_logging
from .utils import read_json, write_json

_VERBOSITY = {0: logging.WARNING, 1: logging.INFO, 2: logging.DEBUG}


def _set_by_path(tree, path, value):
    """path 'a;b;c' -> tree['a']['b']['c'] = value (the `target` syntax of train.py's CustomArgs)."""
    *parents, leaf = path.split(';')
    node = tree
    for key in parents:
        node = node[key]
    node[leaf] = value


def _flag_name(flags):
    long_flags = [f for f in flags if f.startswith('--')]
    return (long_flags[0] if long_flags else flags[0]).replace('--', '')


class ConfigParser:
    def __init__(self, config, resume=None, modification=None, run_id=None):
        for path, value in (modification or {}).items():
            if value is not None:
                _set_by_path(config, path, value)
        self._config = config
        self.resume = resume
        self.log_levels = dict(_VERBOSITY)

        rank, _, _ = parallel.init_distributed(use_cuda=config.get('n_gpu', 0) > 0)
        fresh = run_id is None
        if fresh:
            run_id = parallel.broadcast_object(datetime.now().strftime(r'%m%d_%H%M%S'))
        root = Path(config['trainer']['save_dir'])
        self._save_dir = root / 'models' / config['name'] / run_id
        self._log_dir = root / 'log' / config['name'] / run_id
        # a fresh timestamped run must not reuse a directory; an explicit run_id ('' included) may.  Rank 0 owns the run
        # directory: it creates both directories and writes config.json FIRST, the other ranks wait at the barrier and only
        # then make sure the directories exist (without that order a non-zero rank's mkdir could win the race and rank 0's
        # exist_ok=False would kill the job)
        # ... and if rank 0 fails there (run-id collision, permissions, a full disk) every rank raises the SAME error instead of
        # sitting in a barrier until the process-group timeout: rank 0 broadcasts what happened
        err = None
        if rank == 0:
            try:
                for d in (self._save_dir, self._log_dir):
                    d.mkdir(parents=True, exist_ok=not fresh and run_id == '')
                write_json(config, self._save_dir / 'config.json')
            except Exception as e:      # anything (a non-serialisable config value too): every rank must learn of it
                err = f'{type(e).__name__}: {e}'
        err = parallel.broadcast_object(err)
        if err is not None:
            raise RuntimeError(f'rank 0 could not create the run directory {self._save_dir}: {err}')
        if rank != 0:
            for d in (self._save_dir, self._log_dir):
                d.mkdir(parents=True, exist_ok=True)
        setup_logging(self._log_dir)

    # ------------------------------------------------------------------ construction from the command line
    @classmethod
    def from_args(cls, args, options=''):
        """args: an argparse parser carrying -c/--config, -r/--resume, -d/--device; options: CustomArgs(flags, type, target)."""
        for opt in options:
            args.add_argument(*opt.flags, default=None, type=opt.type)
        if not isinstance(args, tuple):
            args = args.parse_args()
        if args.device is not None:
            os.environ["CUDA_VISIBLE_DEVICES"] = args.device
        resume = Path(args.resume) if args.resume is not None else None
        if resume is None and args.config is None:
            raise AssertionError("Configuration file need to be specified. Add '-c config.json', for example.")
        config = read_json(resume.parent / 'config.json' if resume is not None else Path(args.config))
        if resume is not None and args.config:
            config.update(read_json(args.config))     # fine-tuning: a new config on top of the checkpoint's
        overrides = {opt.target: getattr(args, _flag_name(opt.flags)) for opt in options}
        return cls(config, resume, overrides)

    # ------------------------------------------------------------------ reflection factories
    def _spec(self, name, extra):
        """(type name, constructor kwargs) of config[name]; kwargs given in code may not shadow configured ones."""
        entry = self[name]
        kwargs = dict(entry['args'])
        clash = [k for k in extra if k in kwargs]
        if clash:
            raise AssertionError('Overwriting kwargs given in config file is not allowed')
        kwargs.update(extra)
        return entry['type'], kwargs

    def init_obj(self, name, module, *args, **kwargs):
        """getattr(module, config[name]['type'])(*args, **config[name]['args'], **kwargs)"""
        type_name, ctor_kwargs = self._spec(name, kwargs)
        return getattr(module, type_name)(*args, **ctor_kwargs)

    def init_ftn(self, name, module, *args, **kwargs):
        """Like init_obj, but returns the callable with its arguments bound instead of calling it."""
        type_name, ctor_kwargs = self._spec(name, kwargs)
        return partial(getattr(module, type_name), *args, **ctor_kwargs)

    def restore_snapshot(self, name, module, *args, **kwargs):
        """Build config[name]['type'] and load config[name]['snapshot'] into it (shape-matched partial load)."""
        from . import models
        net = self.init_obj(name, module, *args, **kwargs)
        net, _ = models.load_weights(self[name]['snapshot'], net, None, False)
        return net

    # ------------------------------------------------------------------ access
    def __getitem__(self, name):
        return self._config[name]

    def __contains__(self, name):
        return name in self._config

    def get_logger(self, name, verbosity=2):
        if verbosity not in self.log_levels:
            raise AssertionError('verbosity option {} is invalid. Valid options are {}.'.format(verbosity, self.log_levels.keys()))
        logger = logging.getLogger(name)
        logger.setLevel(self.log_levels[verbosity])
        return logger

    config = property(lambda self: self._config)
    save_dir = property(lambda self: self._save_dir)
    log_dir = property(lambda self: self._log_dir)
