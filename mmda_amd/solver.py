"""Solver with the reference's surface (reference src/solver.py:42-462): ``Solver(train_config, dev_config, test_config,
train_dl, dev_dl, test_dl, is_train, model)``, ``build()``, ``train()``, ``eval(mode)`` and the six ``get_*_loss``
getters - plus ``train_epoch()`` (the body of solver.py:127-197), which is what ``train()`` calls.

Two ways through a batch, same kernels underneath:
  * ``train_epoch()``           - fused native step (``MISA.train_step``): zero_grad, forward, six losses, backward,
                                  [RCCL gradient all-reduce], clip + Adam; losses are read back once per epoch.
  * ``train_epoch_unfused()``   - the reference's statement order verbatim (forward, six getters on the module's
                                  side-channel attributes, ``loss.backward()``, clip, ``optimizer.step()``) through autograd.
wandb / hypertune / tqdm / sklearn reports of the reference are logging only and are not reproduced.
"""
from __future__ import annotations

import os

import numpy as np
import torch

from . import models
from . import optim as _optim
from .dist import DataParallelSync
from .utils import to_gpu, to_cpu, set_device
from .utils import functions as F


from .utils.eval import get_accuracy, get_metrics, DeviceEval      # reference utils/eval.py


class Solver(object):
    def __init__(self, train_config, dev_config, test_config, train_data_loader, dev_data_loader, test_data_loader,
                 is_train=True, model=None):
        self.train_config = train_config
        self.epoch_i = 0
        self.train_data_loader = train_data_loader
        self.dev_data_loader = dev_data_loader
        self.test_data_loader = test_data_loader
        self.is_train = is_train
        self.model = model
        if torch.cuda.is_available():
            self.device = torch.device(train_config.device)
        else:
            self.device = torch.device("cpu")
        set_device(self.device)
        self.dp = None
        self.loss_diff = F.DiffLoss()
        self.loss_cmd = F.CMD()

    # ------------------------------------------------------------------ build (solver.py:60-100)
    def build(self, cuda=True):
        cfg = self.train_config
        if self.model is None:
            self.model = getattr(models, cfg.model)(cfg)
        for name, param in self.model.named_parameters():
            if "weight_hh" in name:
                torch.nn.init.orthogonal_(param)
        if not getattr(cfg, "use_bert", False):
            if getattr(cfg, "pretrained_emb", None) is not None:
                self.model.embed.weight.data = cfg.pretrained_emb
            self.model.embed.requires_grad = False     # a no-op in the reference too: the Parameter keeps training
        self.model.to(self.device)
        if self.is_train:
            self.optimizer = cfg.optimizer([p for p in self.model.parameters() if p.requires_grad], lr=cfg.learning_rate)
            if hasattr(self.optimizer, "attach"):
                self.optimizer.attach(self.model)
        if torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1:
            # config.dp_global_stats (not a reference option: the reference is single-device): the batch-statistic losses on the
            # batch of all ranks instead of DDP semantics (mmda_amd/dist.py)
            self.dp = DataParallelSync(global_stats=bool(getattr(cfg, "dp_global_stats", False)))
            self.dp.broadcast_parameters(self.model)
        return self

    # ------------------------------------------------------------------ train (solver.py:103-307)
    def train_epoch(self):
        """Fused path.  Returns the epoch's mean losses (dict)."""
        cfg = self.train_config
        self.model.train()
        sums = None
        n = 0
        for batch in self.train_data_loader:
            t, v, a, y, emo_label, l, bert_sent, bert_sent_type, bert_sent_mask, ids = batch
            t = to_gpu(t); v = to_gpu(v); a = to_gpu(a); emo_label = to_gpu(emo_label)
            l = to_cpu(l)
            self.model.train_step(t, v, a, l, emo_label, lr=cfg.learning_rate, clip=cfg.clip,
                                  grad_sync=self.dp.sync if self.dp is not None else None,
                                  optimizer=getattr(self, "optimizer", None))
            L = self.model._ws_view("losses", (8,))
            sums = L.clone() if sums is None else sums + L       # stays on the device; one sync per epoch
            n += 1
        out = (sums / max(n, 1)).tolist() if sums is not None else [0.0] * 8
        # the read-back above synchronised anyway: a recurrence that gave up on its cluster invalidates the whole epoch
        self._check_cluster("train_epoch")
        return dict(cls=out[0], diff=out[1], sim=out[2], recon=out[3], conf=out[4], total=out[5])

    def train_epoch_unfused(self):
        """The reference's loop body, statement for statement (solver.py:138-193), through autograd."""
        cfg = self.train_config
        self.model.train()
        train_loss = []
        for batch in self.train_data_loader:
            self.model.zero_grad()
            t, v, a, y, emo_label, l, bert_sent, bert_sent_type, bert_sent_mask, ids = batch
            t = to_gpu(t); v = to_gpu(v); a = to_gpu(a); y = to_gpu(y); emo_label = to_gpu(emo_label)
            l = to_cpu(l)
            predicted_scores, predicted_labels = self.model(t, v, a, l, bert_sent, bert_sent_type, bert_sent_mask)
            emo_label = emo_label.type(torch.float)
            cls_loss = self.get_cls_loss(predicted_scores, emo_label)
            diff_loss = self.get_diff_loss()
            domain_loss = self.get_domain_loss()
            recon_loss = self.get_recon_loss()
            cmd_loss = self.get_cmd_loss()
            conf_loss = self.get_conf_loss(predicted_scores, emo_label)
            similarity_loss = cmd_loss if cfg.use_cmd_sim else domain_loss
            loss = cls_loss + cfg.diff_weight * diff_loss + cfg.sim_weight * similarity_loss + cfg.recon_weight * recon_loss
            if cfg.use_confidNet:
                loss = loss + cfg.conf_weight * conf_loss
            loss.backward()
            _optim.clip_grad_value_(self.model, cfg.clip)
            self.optimizer.step()
            train_loss.append(loss.item())
        self._check_cluster("train_epoch_unfused")
        return dict(total=float(np.mean(train_loss)) if train_loss else 0.0)

    def _check_cluster(self, where):
        if hasattr(self.model, "check_cluster"):
            self.model.check_cluster(where)

    def _check_cluster_all_ranks(self, where):
        """Data parallel: every rank reads its own status, the ranks agree (one MAX all-reduce) and ALL of them raise -- a rank
        that raised alone would leave its peers waiting in the next collective."""
        if self.dp is None:
            return self._check_cluster(where)
        bad = bool(self.model.cluster_aborted()) if hasattr(self.model, "cluster_aborted") else False
        if self.dp.any_rank(bad):
            from . import _lib
            raise _lib.MMDAError(f"a resident-weights recurrence timed out waiting for its workgroup cluster on "
                                 f"{'this rank' if bad else 'another rank'} ({where}): results since then are invalid")

    def train(self):
        cfg = self.train_config
        best_valid_loss = float("inf")
        best_epoch = -1
        history = []
        for e in range(cfg.n_epoch):
            self.epoch_i = e
            tr = self.train_epoch()
            print(f"Training loss: {round(tr['total'], 4)}")
            valid_loss, valid_acc, preds, truths = self.eval(mode="dev")
            print("-" * 100)
            print("Epochs: {}, Valid loss: {}, Valid acc: {}".format(e, valid_loss, valid_acc))
            print("-" * 100)
            # Data parallel: whether this epoch is the best one is decided ONCE, from rank 0's dev loss (a sharded dev loader, or one
            # ULP of difference between ranks, must not send one rank into the checkpoint barrier and another into the next epoch's
            # all-reduce), and the cluster check in front of the save is collective, so an error is raised on every rank.
            decided = self.dp.agree(valid_loss) if self.dp is not None else valid_loss
            if decided <= best_valid_loss:
                best_valid_loss, best_epoch = decided, e
                print("Found new best model on dev set!")
                self._check_cluster_all_ranks("before saving the checkpoint")  # never save weights a failed exchange produced
                if self.dp is None or self.dp.rank == 0:
                    os.makedirs("checkpoints", exist_ok=True)
                    torch.save(self.model.state_dict(), f"checkpoints/model_{cfg.name}.std")
                    torch.save(self.optimizer.state_dict(), f"checkpoints/optim_{cfg.name}.std")     # solver.py:220
            if self.dp is not None:
                torch.distributed.barrier()               # every epoch, on every rank: nobody reads the checkpoint before rank 0 has written it
            history.append(dict(epoch=e, train=tr, valid_loss=valid_loss, valid_acc=valid_acc))
        test_loss, acc, _, _ = self.eval(mode="test", to_print=best_epoch >= 0)
        print("=" * 50)
        print(f"Best epoch: {best_epoch}")
        print(f"Accuracy: {acc}")
        return history

    # ------------------------------------------------------------------ eval (solver.py:311-370)
    def eval(self, mode=None, to_print=False):
        assert mode is not None
        self.model.eval()
        y_true, y_pred, eval_loss = [], [], []
        dataloader = self.dev_data_loader if mode == "dev" else self.test_data_loader
        if mode == "test" and to_print:
            path = f"checkpoints/model_{self.train_config.name}.std"
            if os.path.exists(path):
                self.model.load_state_dict(torch.load(path, weights_only=True))
        # Device-side evaluation: predictions, the per-batch classification loss and the metric counts stay on the GPU; one
        # read-back at the end of the pass instead of three per batch (solver.py:350-360 calls .item()/.cpu() per batch).
        acc_dev = None
        with torch.no_grad():
            for batch in dataloader:
                t, v, a, y, emo_label, l, bert_sent, bert_sent_type, bert_sent_mask, ids = batch
                t = to_gpu(t); v = to_gpu(v); a = to_gpu(a); emo_label = to_gpu(emo_label)
                l = to_cpu(l)
                predicted_scores, predicted_labels = self.model(t, v, a, l, bert_sent, bert_sent_type, bert_sent_mask)
                emo_label = emo_label.type(torch.float)
                if predicted_labels.is_cuda:
                    if acc_dev is None:
                        acc_dev = DeviceEval(predicted_labels.shape[1], predicted_labels.device)
                    acc_dev.update(predicted_labels, emo_label)
                    acc_dev.add_cls_loss(predicted_scores, emo_label)
                else:
                    eval_loss.append(self.get_cls_loss(predicted_scores, emo_label).item())
                y_pred.append(predicted_labels.detach())
                y_true.append(emo_label.detach())
        y_true = torch.cat(y_true, 0).cpu().numpy().squeeze() if y_true else np.zeros((0,))
        y_pred = torch.cat(y_pred, 0).cpu().numpy().squeeze() if y_pred else np.zeros((0,))
        self._check_cluster(f"eval({mode})")
        if acc_dev is not None:
            loss, acc, self.last_eval_metrics = acc_dev.result()
            return loss, acc, y_pred, y_true
        eval_loss = float(np.mean(eval_loss)) if eval_loss else 0.0
        self.last_eval_metrics = get_metrics(y_true, y_pred)
        return eval_loss, get_accuracy(y_true, y_pred), y_pred, y_true

    # ------------------------------------------------------------------ getters (solver.py:373-462)
    def get_cls_loss(self, predicted_scores, emo_label):
        return F.bce_sum_over_classes(predicted_scores, emo_label.type(torch.float))

    def get_domain_loss(self):
        if self.train_config.use_cmd_sim:
            return 0.0
        m = self.model
        return F.domain_loss(m.domain_label_t, m.domain_label_v, m.domain_label_a)

    def get_cmd_loss(self):
        if not self.train_config.use_cmd_sim:
            return 0.0
        m = self.model
        # (t,v) + (t,a) + (a,v), /3  - one fused launch instead of three CMD() calls
        return F.cmd_loss_multi([m.utt_shared_t, m.utt_shared_v, m.utt_shared_a], [(0, 1), (0, 2), (2, 1)], 5, 1.0 / 3.0)

    def get_diff_loss(self):
        m = self.model
        ts = [m.utt_private_t, m.utt_private_v, m.utt_private_a, m.utt_shared_t, m.utt_shared_v, m.utt_shared_a]
        # (p_t,s_t) (p_v,s_v) (p_a,s_a) (p_a,p_t) (p_a,p_v) (p_t,p_v)   solver.py:432-439
        return F.diff_loss_multi(ts, [(0, 3), (1, 4), (2, 5), (2, 0), (2, 1), (0, 1)])

    def get_recon_loss(self):
        m = self.model
        return F.recon_loss([m.utt_t_recon, m.utt_v_recon, m.utt_a_recon], [m.utt_t_orig, m.utt_v_orig, m.utt_a_orig])

    def get_conf_loss(self, pred, truth):
        return F.conf_loss(pred, self.model.tcp, truth)
