"""Epoch driver and checkpoint I/O behind the kept entrypoints (the role of v2/base/base_trainer.py's
``Multi_BaseTrainer_dist``), written for the HIP step engine.

Only the EXTERNAL CONTRACT is the reference's -- everything a kept script, config file or old checkpoint can observe:
  * the constructor signature and the ``config['trainer']`` keys (``epochs``, ``save_period``, ``verbosity``, ``monitor``,
    ``init_val``, ``early_stop``);
  * the monitor rule ``"min <key>" / "max <key>" / "off"`` with ties counting as improvements (v2/base/base_trainer.py:117-136);
    the reference counts epochs without improvement but its early-stop ``break`` is commented out (:138-141), so training
    always runs to ``epochs`` here too;
  * the epoch log: metric lists keyed by the metric functions' names, ``val_`` prefixed ones, and the nested validation
    dict flattened to ``val_<loader>_<metric>_<statistic>`` (:96-107);
  * the checkpoint: rank 0 writes ``{'arch','epoch','state_dict','optimizer','monitor_best','config'}`` to
    ``<save_dir>/checkpoint-epoch{N}.pth`` and ``model_best.pth`` (:165-189); resume restores epoch, best value, weights
    (undoing a ``module.`` prefix) and -- when the optimizer type is unchanged -- the optimizer state (:191-247).
The structure is this package's own: a ``Monitor`` that owns the improvement rule, a recursive ``flatten_epoch_log``,
and checkpoint helpers that v1's extra state (the dropout seed) plugs into.  There is no DistributedDataParallel wrap:
gradients are averaged by tvts_amd.dist.GradSync inside the step.
"""
from __future__ import annotations

import math
import os
from typing import Callable, Dict, Iterable, Mapping, Optional

import torch


class Monitor:
    """The ``config['trainer']['monitor']`` rule: which logged value decides that an epoch is the best so far."""

    def __init__(self, spec: str = "off"):
        self.mode, self.key = "off", None
        self.best = 0
        self.stale_epochs = 0
        if spec != "off":
            mode, key = spec.split()
            if mode not in ("min", "max"):
                raise AssertionError(f"monitor mode must be 'min' or 'max', got {mode!r}")
            self.mode, self.key = mode, key
            self.best = math.inf if mode == "min" else -math.inf

    @property
    def active(self) -> bool:
        return self.mode != "off"

    def observe(self, log: Mapping, warn: Callable[[str], None]) -> bool:
        """True when ``log`` holds a new best value of the monitored key (an equal value counts).  A missing key switches the
        monitor off for the rest of the run, with the reference's warning."""
        if not self.active:
            return False
        if self.key not in log:
            warn("Warning: Metric '{}' is not found. Model performance monitoring is disabled.".format(self.key))
            self.mode = "off"
            return False
        value = log[self.key]
        better = value <= self.best if self.mode == "min" else value >= self.best
        if better:
            self.best, self.stale_epochs = value, 0
        else:
            self.stale_epochs += 1
        return better


def flatten_epoch_log(epoch: int, result: Mapping, metric_names: Iterable[str]) -> Dict[str, object]:
    """One flat ``{name: value}`` dict per epoch from what ``_train_epoch`` returned."""
    names = list(metric_names)
    flat: Dict[str, object] = {"epoch": epoch}

    def spread(prefix: str, node):
        if isinstance(node, Mapping):
            for k, v in node.items():
                spread(f"{prefix}_{k}", v)
        else:
            flat[prefix] = node

    for key, value in result.items():
        if key == "metrics":
            flat.update(zip(names, value))
        elif key == "val_metrics":
            flat.update(zip(("val_" + n for n in names), value))
        elif key == "nested_val_metrics":
            spread("val", value)
        else:
            flat[key] = value
    return flat


def strip_data_parallel_prefix(state_dict: Mapping) -> Mapping:
    """Checkpoints written from a (Distributed)DataParallel wrap carry ``module.`` in front of every key; this model is never
    wrapped, so only that direction exists (v2/utils/util.py:25-50 handles both)."""
    first = next(iter(state_dict), "")
    if first.startswith("module."):
        return {k[len("module."):]: v for k, v in state_dict.items()}
    return state_dict


class Multi_BaseTrainer_dist:
    CHECKPOINT_NAME = "checkpoint-epoch{}.pth"
    BEST_NAME = "model_best.pth"

    def __init__(self, args, model, loss, metrics, optimizer, config, writer=None, init_val=False):
        trainer_cfg = config["trainer"]
        self.args, self.config, self.writer = args, config, writer
        self.logger = config.get_logger("trainer", trainer_cfg["verbosity"])
        self.model = model
        self.device = self.model.device = model.store.device
        self.loss = loss.to(self.device) if hasattr(loss, "to") else loss
        self.metrics, self.optimizer = metrics, optimizer
        self.epochs, self.save_period = trainer_cfg["epochs"], trainer_cfg["save_period"]
        self.init_val = trainer_cfg.get("init_val", True)
        self._monitor = Monitor(trainer_cfg.get("monitor", "off"))
        if self._monitor.active:
            self.early_stop = trainer_cfg.get("early_stop", math.inf)
        self.start_epoch = 1
        self.checkpoint_dir = config.save_dir
        resume = getattr(config, "resume", None)
        if resume is not None:
            self._resume_checkpoint(resume)

    # the reference's attribute names, for scripts and tests that read them
    monitor = property(lambda self: "off" if not self._monitor.active else f"{self._monitor.mode} {self._monitor.key}")
    mnt_mode = property(lambda self: self._monitor.mode)
    mnt_metric = property(lambda self: self._monitor.key)

    @property
    def mnt_best(self):
        return self._monitor.best

    @mnt_best.setter
    def mnt_best(self, value):
        self._monitor.best = value

    def _train_epoch(self, epoch):
        raise NotImplementedError

    # ------------------------------------------------------------------ epochs
    def train(self):
        lead = self.args.rank == 0  # rank 0 alone logs, monitors and writes checkpoints
        if self.init_val and getattr(self, "do_validation", False):
            self._valid_epoch(-1)
        for epoch in range(self.start_epoch, self.epochs + 1):
            result = self._train_epoch(epoch)
            if not lead:
                continue
            log = flatten_epoch_log(epoch, result, (m.__name__ for m in self.metrics))
            for name, value in log.items():
                self.logger.info("    {:15s}: {}".format(str(name), value))
            is_best = self._monitor.observe(log, self.logger.warning)
            if is_best or epoch % self.save_period == 0:
                self._save_checkpoint(epoch, save_best=is_best)

    # ------------------------------------------------------------------ checkpoints
    def _extra_state(self) -> Optional[dict]:
        """Trainer-specific state that must survive a resume beside the reference's keys (v1: the dropout seed)."""
        return None

    def _load_extra_state(self, state: Optional[dict]) -> None:
        pass

    def _save_checkpoint(self, epoch, save_best=False):
        payload = dict(arch=type(self.model).__name__, epoch=epoch, state_dict=self.model.state_dict(),
                       optimizer=self.optimizer.state_dict(), monitor_best=self._monitor.best, config=self.config)
        extra = self._extra_state()
        if extra:
            payload["tvts_amd"] = extra  # ignored by the reference's loader, which reads its six keys by name
        folder = str(self.checkpoint_dir)
        path = os.path.join(folder, self.CHECKPOINT_NAME.format(epoch))
        torch.save(payload, path)
        self.logger.info("Saving checkpoint: {} ...".format(path))
        if save_best:
            torch.save(payload, os.path.join(folder, self.BEST_NAME))
            self.logger.info("Saving current best: model_best.pth ...")

    def _resume_checkpoint(self, resume_path):
        path = str(resume_path)
        self.logger.info("Loading checkpoint: {} ...".format(path))
        ckpt = torch.load(path, map_location=self.device, weights_only=False)  # the dict carries the ConfigParser object
        saved_cfg = ckpt["config"]
        if saved_cfg["arch"] != self.config["arch"]:
            self.logger.warning("Warning: Architecture configuration given in config file is different from that of "
                                "checkpoint. This may yield an exception while state_dict is being loaded.")
        self.model.load_state_dict(strip_data_parallel_prefix(ckpt["state_dict"]))
        if saved_cfg["optimizer"]["type"] == self.config["optimizer"]["type"]:
            self.optimizer.load_state_dict(ckpt["optimizer"])  # a failing load is an error, as in the reference
        else:
            self.logger.warning("Warning: Optimizer type given in config file is different from that of checkpoint. "
                                "Optimizer parameters not being resumed.")
        self._load_extra_state(ckpt.get("tvts_amd"))
        self._monitor.best = ckpt["monitor_best"]
        self.start_epoch = ckpt["epoch"] + 1
        self.logger.info("Checkpoint loaded. Resume training from epoch {}".format(self.start_epoch))
