"""`ConfigParser`: JSON experiment config -> objects, with the reference's interface (parse_config.py:11-175):
`config[name]`, `init_obj(name, module, *a, **kw)`, `init_ftn`, `restore_snapshot(name, module)`, `get_logger`,
`save_dir` / `log_dir`, `from_args`.  The JSON schema (teacher, *_data_loader, optimizer, supervised_loss, kd_loss,
hint_loss, lr_scheduler, trainer, pruning{args,pruning_plan,hint,unfreeze}, weight_scheduler, ...) is unchanged."""
import logging
import os
from datetime import datetime
from functools import partial, reduce
from operator import getitem
from pathlib import Path

from .logger import setup_logging
from .utils import read_json, write_json


class ConfigParser:
    def __init__(self, config, resume=None, modification=None, run_id=None):
        self._config = _update_config(config, modification)
        self.resume = resume
        save_dir = Path(self.config['trainer']['save_dir'])
        exper_name = self.config['name']
        if run_id is None:
            run_id = datetime.now().strftime(r'%m%d_%H%M%S')
        self._save_dir = save_dir / 'models' / exper_name / run_id
        self._log_dir = save_dir / 'log' / exper_name / run_id
        exist_ok = run_id == ''
        self.save_dir.mkdir(parents=True, exist_ok=exist_ok)
        self.log_dir.mkdir(parents=True, exist_ok=exist_ok)
        write_json(self.config, self.save_dir / 'config.json')
        setup_logging(self.log_dir)
        self.log_levels = {0: logging.WARNING, 1: logging.INFO, 2: logging.DEBUG}

    @classmethod
    def from_args(cls, args, options=''):
        for opt in options:
            args.add_argument(*opt.flags, default=None, type=opt.type)
        if not isinstance(args, tuple):
            args = args.parse_args()
        if args.device is not None:
            os.environ["CUDA_VISIBLE_DEVICES"] = args.device
        if args.resume is not None:
            resume = Path(args.resume)
            cfg_fname = resume.parent / 'config.json'
        else:
            assert args.config is not None, "Configuration file need to be specified. Add '-c config.json', for example."
            resume = None
            cfg_fname = Path(args.config)
        config = read_json(cfg_fname)
        if args.config and resume:
            config.update(read_json(args.config))
        modification = {opt.target: getattr(args, _get_opt_name(opt.flags)) for opt in options}
        return cls(config, resume, modification)

    def _type_args(self, name, kwargs):
        module_args = dict(self[name]['args'])
        assert all([k not in module_args for k in kwargs]), 'Overwriting kwargs given in config file is not allowed'
        module_args.update(kwargs)
        return self[name]['type'], module_args

    def init_obj(self, name, module, *args, **kwargs):
        """config.init_obj('name', module, a, b=1)  ==  getattr(module, config['name']['type'])(a, **config['name']['args'], b=1)"""
        t, module_args = self._type_args(name, kwargs)
        return getattr(module, t)(*args, **module_args)

    def init_ftn(self, name, module, *args, **kwargs):
        t, module_args = self._type_args(name, kwargs)
        return partial(getattr(module, t), *args, **module_args)

    def restore_snapshot(self, name, module, *args, **kwargs):
        """Build config[name]['type'] from `module` and load config[name]['snapshot'] into it (forgiving restore)."""
        t, module_args = self._type_args(name, kwargs)
        net = getattr(module, t)(*args, **module_args)
        from . import models
        net, _ = models.load_weights(self[name]['snapshot'], net, None, False)
        return net

    def __getitem__(self, name):
        return self.config[name]

    def __contains__(self, name):
        return name in self.config

    def get_logger(self, name, verbosity=2):
        assert verbosity in self.log_levels, 'verbosity option {} is invalid. Valid options are {}.'.format(
            verbosity, self.log_levels.keys())
        logger = logging.getLogger(name)
        logger.setLevel(self.log_levels[verbosity])
        return logger

    @property
    def config(self):
        return self._config

    @property
    def save_dir(self):
        return self._save_dir

    @property
    def log_dir(self):
        return self._log_dir


def _update_config(config, modification):
    if modification is None:
        return config
    for k, v in modification.items():
        if v is not None:
            keys = k.split(';')
            getitem_by = reduce(getitem, keys[:-1], config)
            getitem_by[keys[-1]] = v
    return config


def _get_opt_name(flags):
    for flg in flags:
        if flg.startswith('--'):
            return flg.replace('--', '')
    return flags[0].replace('--', '')
