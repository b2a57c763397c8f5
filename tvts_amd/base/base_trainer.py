"""Epoch driver + checkpoint layout of v2/base/base_trainer.py (Multi_BaseTrainer_dist), rebuilt for the
HIP step engine.  What is kept bit-for-bit is the external contract: constructor arguments, the
``config['trainer']`` keys, the monitor / early-stop rule (:117-136), the checkpoint dict
``{'arch','epoch','state_dict','optimizer','monitor_best','config'}`` written by rank 0 to
``<save_dir>/checkpoint-epoch{N}.pth`` / ``model_best.pth`` (:165-189) and resume with the ``module.``
prefix fix (:191-247).  There is no DistributedDataParallel wrap: gradients are averaged by
tvts_amd.dist.GradSync inside the step.
"""
from __future__ import annotations

import os
from abc import abstractmethod

import torch
from numpy import inf


class Multi_BaseTrainer_dist:
    def __init__(self, args, model, loss, metrics, optimizer, config, writer=None, init_val=False):
        self.config = config
        self.logger = config.get_logger("trainer", config["trainer"]["verbosity"])
        self.init_val = init_val
        self.args = args
        self.device = model.store.device
        self.model = model
        self.model.device = self.device
        self.loss = loss.to(self.device) if hasattr(loss, "to") else loss
        self.metrics = metrics
        self.optimizer = optimizer
        cfg = config["trainer"]
        self.epochs = cfg["epochs"]
        self.save_period = cfg["save_period"]
        self.monitor = cfg.get("monitor", "off")
        self.init_val = cfg.get("init_val", True)
        if self.monitor == "off":
            self.mnt_mode, self.mnt_best = "off", 0
        else:
            self.mnt_mode, self.mnt_metric = self.monitor.split()
            assert self.mnt_mode in ["min", "max"]
            self.mnt_best = inf if self.mnt_mode == "min" else -inf
            self.early_stop = cfg.get("early_stop", inf)
        self.start_epoch = 1
        self.checkpoint_dir = config.save_dir
        self.writer = writer
        if getattr(config, "resume", None) is not None:
            self._resume_checkpoint(config.resume)

    @abstractmethod
    def _train_epoch(self, epoch):
        raise NotImplementedError

    def train(self):
        not_improved_count = 0
        if self.init_val and getattr(self, "do_validation", False):
            self._valid_epoch(-1)
        for epoch in range(self.start_epoch, self.epochs + 1):
            result = self._train_epoch(epoch)
            log = {"epoch": epoch}
            for key, value in result.items():
                if key == "metrics":
                    log.update({mtr.__name__: value[i] for i, mtr in enumerate(self.metrics)})
                elif key == "val_metrics":
                    log.update({"val_" + mtr.__name__: value[i] for i, mtr in enumerate(self.metrics)})
                else:
                    log[key] = value
            if self.args.rank == 0:
                for key, value in log.items():
                    self.logger.info("    {:15s}: {}".format(str(key), value))
            best = False
            if self.mnt_mode != "off" and self.args.rank == 0:
                try:
                    improved = (self.mnt_mode == "min" and log[self.mnt_metric] <= self.mnt_best) or \
                               (self.mnt_mode == "max" and log[self.mnt_metric] >= self.mnt_best)
                except KeyError:
                    self.logger.warning("Warning: Metric '{}' is not found. Model performance monitoring is "
                                        "disabled.".format(self.mnt_metric))
                    self.mnt_mode, improved, not_improved_count = "off", False, 0
                if improved:
                    self.mnt_best, not_improved_count, best = log[self.mnt_metric], 0, True
                else:
                    not_improved_count += 1
                if not_improved_count > self.early_stop:
                    self.logger.info("Validation performance didn't improve for {} epochs. Training stops.".format(
                        self.early_stop))
                    break
            if self.args.rank == 0 and (epoch % self.save_period == 0 or best):
                self._save_checkpoint(epoch, save_best=best)

    def _save_checkpoint(self, epoch, save_best=False):
        state = {
            "arch": type(self.model).__name__,
            "epoch": epoch,
            "state_dict": self.model.state_dict(),
            "optimizer": self.optimizer.state_dict(),
            "monitor_best": self.mnt_best,
            "config": self.config,
        }
        filename = str(os.path.join(str(self.checkpoint_dir), "checkpoint-epoch{}.pth".format(epoch)))
        torch.save(state, filename)
        self.logger.info("Saving checkpoint: {} ...".format(filename))
        if save_best:
            best_path = str(os.path.join(str(self.checkpoint_dir), "model_best.pth"))
            torch.save(state, best_path)
            self.logger.info("Saving current best: model_best.pth ...")

    def _resume_checkpoint(self, resume_path):
        resume_path = str(resume_path)
        self.logger.info("Loading checkpoint: {} ...".format(resume_path))
        checkpoint = torch.load(resume_path, map_location=self.device, weights_only=False)
        self.start_epoch = checkpoint["epoch"] + 1
        self.mnt_best = checkpoint["monitor_best"]
        sd = checkpoint["state_dict"]
        if next(iter(sd)).startswith("module."):
            sd = {k[7:]: v for k, v in sd.items()}
        self.model.load_state_dict(sd)
        try:
            self.optimizer.load_state_dict(checkpoint["optimizer"])
        except Exception as e:  # different optimizer type: same rule as the reference (:241-245)
            self.logger.warning("Warning: optimizer state not restored ({}).".format(e))
        self.logger.info("Checkpoint loaded. Resume training from epoch {}".format(self.start_epoch))
