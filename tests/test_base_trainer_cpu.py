"""Epoch-driver contract of v2/base/base_trainer.py (Multi_BaseTrainer_dist.train) that needs no GPU: the epoch log
(including the `nested_val_metrics` flattening, :100-105), the monitor rule without early stopping (:117-141) and the
checkpoint cadence (:143-146)."""
import logging
import types

import pytest
import torch


class _Cfg(dict):
    resume = None

    def __init__(self, save_dir, **trainer):
        super().__init__(trainer=dict(dict(epochs=5, save_period=2, verbosity=2, monitor="max val_score", early_stop=1,
                                           init_val=False), **trainer),
                         arch=dict(type="X"), optimizer=dict(type="AdamW"))
        self.save_dir = save_dir

    def get_logger(self, name, verbosity=2):
        return logging.getLogger(name)


def _trainer(tmp_path, results, **trainer):
    from tvts_amd.base.base_trainer import Multi_BaseTrainer_dist

    class T(Multi_BaseTrainer_dist):
        saved = []

        def _train_epoch(self, epoch):
            return results[epoch - 1]

        def _save_checkpoint(self, epoch, save_best=False):
            self.saved.append((epoch, save_best))
    model = types.SimpleNamespace(store=types.SimpleNamespace(device=torch.device("cpu")))
    args = types.SimpleNamespace(rank=0, local_rank=0, world_size=1)
    t = T(args, model, loss=object(), metrics=[], optimizer=None, config=_Cfg(str(tmp_path), **trainer))
    t.saved = []
    return t


def test_epoch_log_flattens_nested_val_metrics_and_never_stops_early(tmp_path, caplog):
    nested = {0: {"t2v_metrics": {"R1": 12.5, "MedR": 3.0}}}
    results = [dict(loss_0=1.0, val_score=0.5, nested_val_metrics=nested)] + [dict(loss_0=1.0, val_score=0.1)] * 4
    t = _trainer(tmp_path, results)
    with caplog.at_level(logging.INFO):
        t.train()
    text = caplog.text
    assert "val_0_t2v_metrics_R1" in text and "12.5" in text and "val_0_t2v_metrics_MedR" in text
    # 4 epochs without improvement with early_stop = 1: the reference's break is commented out -> all 5 epochs ran
    assert [e for e, _ in t.saved] == [1, 2, 4] and t.saved[0] == (1, True)
    assert t.mnt_best == 0.5


def test_missing_monitor_metric_disables_monitoring(tmp_path):
    t = _trainer(tmp_path, [dict(loss_0=1.0)] * 2, epochs=2, save_period=1)
    t.train()
    assert t.mnt_mode == "off" and t.saved == [(1, False), (2, False)]
