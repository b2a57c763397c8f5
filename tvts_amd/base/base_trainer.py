"""Epoch driver + checkpoint layout of v2/base/base_trainer.py (Multi_BaseTrainer_dist), rebuilt for the
HIP step engine.  What is kept bit-for-bit is the external contract: constructor arguments, the
``config['trainer']`` keys, the monitor rule (:117-136; early stopping is commented out there and absent here), the
``nested_val_metrics`` flattening of the epoch log (:100-105), the checkpoint dict
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
        if self.init_val and getattr(self, "do_validation", False):  # (:86-87; the reference's configs always validate)
            self._valid_epoch(-1)
        for epoch in range(self.start_epoch, self.epochs + 1):
            result = self._train_epoch(epoch)
            log = {"epoch": epoch}
            for key, value in result.items():
                if self.args.rank != 0:  # only rank 0 fills the log (:95)
                    continue
                if key == "metrics":
                    log.update({mtr.__name__: value[i] for i, mtr in enumerate(self.metrics)})
                elif key == "val_metrics":
                    log.update({"val_" + mtr.__name__: value[i] for i, mtr in enumerate(self.metrics)})
                elif key == "nested_val_metrics":  # two layers of nesting, flattened into the epoch log (:100-105)
                    for subkey, subval in value.items():
                        for subsubkey, subsubval in subval.items():
                            for subsubsubkey, subsubsubval in subsubval.items():
                                log[f"val_{subkey}_{subsubkey}_{subsubsubkey}"] = subsubsubval
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
                    self.mnt_mode, improved = "off", False
                if improved:
                    self.mnt_best, not_improved_count, best = log[self.mnt_metric], 0, True
                else:
                    not_improved_count += 1
                # the reference counts the epochs without improvement but its early-stop break is commented out
                # (:138-141): training always runs to `epochs`, and so does this trainer
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
        if checkpoint["config"]["arch"] != self.config["arch"]:  # (:205-207)
            self.logger.warning("Warning: Architecture configuration given in config file is different from that of "
                                "checkpoint. This may yield an exception while state_dict is being loaded.")
        sd = checkpoint["state_dict"]
        if next(iter(sd)).startswith("module."):  # this model is never DataParallel-wrapped: only the undo case exists
            sd = {k[7:]: v for k, v in sd.items()}
        self.model.load_state_dict(sd)
        # optimizer state only when the optimizer type is unchanged (:241-245); a failing load is an error, as there
        if checkpoint["config"]["optimizer"]["type"] != self.config["optimizer"]["type"]:
            self.logger.warning("Warning: Optimizer type given in config file is different from that of checkpoint. "
                                "Optimizer parameters not being resumed.")
        else:
            self.optimizer.load_state_dict(checkpoint["optimizer"])
        self.logger.info("Checkpoint loaded. Resume training from epoch {}".format(self.start_epoch))
