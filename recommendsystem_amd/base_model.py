"""Train/eval driver with the surface of the reference's model/utils/BaseModel.py.

Same constructor arguments, attribute names and method names (fit, train_epoch, train_step,
eval_step, evaluate, checkpoint_and_earlystop, lr_decay, save_weights, load_weights,
features_to_divice, labels_to_device, regularization_loss), so model scripts written against the
reference run on the HIP layers.  Differences, all deliberate:

* the checkpoint root is a constructor/environment setting instead of a hard-coded home directory
  (BaseModel.py:49);
* evaluate() computes AUC / logloss itself (the reference's sklearn call uses a keyword that current
  sklearn rejects, BaseModel.py:170);
* ``np.float("inf")`` (BaseModel.py:105) is written ``float("inf")``;
* ``fix_earlystop=False`` keeps the reference's comparison (BaseModel.py:176 treats a LOWER AUC as an
  improvement); pass True for the conventional direction;
* ``use_fused_step()`` switches train_step to the fused engine (sparse touched-rows update, one HIP
  graph per step) -- see engine.py; ``use_sparse_step()`` gives every model kind the touched-rows table update
  (sparse_trainer.py).
"""
from __future__ import annotations

import datetime
import logging
import os
import time

import torch
import torch.nn as nn
import torch.nn.functional as F
import torch.optim as optim

from .layers import Embedding_layer


def get_device(gpu=-1):
    """model/utils/utils.py:91-98."""
    if gpu >= 0 and torch.cuda.is_available():
        return torch.device("cuda:" + str(gpu))
    return torch.device("cpu")


def random_all(seed):
    """model/utils/utils.py:69-76."""
    import random
    import numpy as np
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed(seed)


def roc_auc(labels: torch.Tensor, scores: torch.Tensor) -> float:
    """Rank-based ROC AUC with average ranks over ties (what sklearn.roc_auc_score returns)."""
    y = labels.double().flatten()
    s = scores.double().flatten()
    order = torch.argsort(s)
    _, inv, counts = torch.unique_consecutive(s[order], return_inverse=True, return_counts=True)
    ends = torch.cumsum(counts, 0).double()
    avg = (ends - counts.double() + 1 + ends) / 2
    ranks = torch.empty_like(s)
    ranks[order] = avg[inv]
    n_pos = y.sum()
    n_neg = y.numel() - n_pos
    return float((ranks[y == 1].sum() - n_pos * (n_pos + 1) / 2) / (n_pos * n_neg))


def log_loss(labels: torch.Tensor, prob: torch.Tensor, eps: float = 1e-7) -> float:
    p = prob.double().clamp(eps, 1 - eps)
    y = labels.double()
    return float(-(y * p.log() + (1 - y) * (1 - p).log()).mean())


class BaseModel(nn.Module):
    def __init__(self, model_name, dataRecorder, dataset_name="Criteo_x1", patience=2, num_epochs=30,
                 learning_rate=0.001, criterion_type="BCE", optimizer_type="Adam", weight_decay=0, record=False,
                 gpu=-1, emb_regular=None, net_regular=None, checkpoint_root=None, fix_earlystop=False):
        super().__init__()
        self.start_time = time.time()
        self.logger = logging.getLogger("my_logger")
        self.device = get_device(gpu)
        self.dataRecorder = dataRecorder
        self.dataset_name = dataset_name
        self.input_dim = self.dataRecorder.input_dim
        self.num_epochs = num_epochs
        self.learning_rate = learning_rate
        self.model_name = model_name
        self.criterion_type = criterion_type
        self.optimizer_type = optimizer_type
        self.optimizer = None
        self.criterion = None
        self.embedding_layer = Embedding_layer(self.dataRecorder)
        root = checkpoint_root or os.environ.get("RS_CHECKPOINT_ROOT", os.path.join(os.getcwd(), "result"))
        now = datetime.datetime.now()
        self.checkpoint = os.path.join(root, self.dataset_name, now.strftime("%Y%m%d"), now.strftime("%H%M%S"))
        self.stop_training = False
        self.record = record
        self.hidden_dim_list = []
        self.dropout_rate = 0
        self.normalization = None
        self.activation_func = "ReLU"
        self.patience = patience
        self.reduce_lr_on_plateau = True
        self.emb_regular = emb_regular
        self.net_regular = net_regular
        self.fix_earlystop = fix_earlystop
        self.counter = 0
        self.epoch_index = 0
        self._engine = None

    # ---- factories (BaseModel.py:64-77) ------------------------------------------------------
    def initialize_optimizer(self):
        if self.optimizer_type == "SGD":
            self.optimizer = optim.SGD(self.parameters(), lr=self.learning_rate)
        elif self.optimizer_type == "Adam":
            self.optimizer = optim.Adam(self.parameters(), lr=self.learning_rate)
        elif self.optimizer_type == "AdamW":
            self.optimizer = optim.AdamW(self.parameters(), lr=self.learning_rate)
        else:
            raise ValueError("Invalid value for optimizer")

    def initialize_criterion(self):
        if self.criterion_type == "BCE":
            self.criterion = F.binary_cross_entropy
        else:
            raise ValueError("Invalid value for criterion")

    def reset_parameters(self):
        """BaseModel.py:79-91: Xavier-normal nn.Linear weights, zero biases, then custom hooks."""
        def reset_default_params(m):
            # exact types as in the reference, plus this package's HIP-backed nn.Linear subclass
            if type(m) in [nn.Linear, nn.Conv1d] or (isinstance(m, nn.Linear) and type(m).__name__ == "Linear"):
                nn.init.xavier_normal_(m.weight)
                if m.bias is not None:
                    m.bias.data.fill_(0)

        def reset_custom_params(m):
            if hasattr(m, "reset_custom_params"):
                m.reset_custom_params()
        self.apply(reset_default_params)
        self.apply(reset_custom_params)

    def model_to_device(self):
        self.to(device=self.device)
        self._number_dropouts()

    def _number_dropouts(self):
        """The mask stream of a dropout layer is keyed by (seed, salt, step): salts follow the layer's position in THIS
        model, so that two models built alike draw the same masks from the same seed however many other models the
        process built before (every model's constructor ends in model_to_device, as the reference's do,
        model/DeepFM/DeepFM.py:40-41)."""
        from .layers import Dropout
        k = 0
        for mod in self.modules():
            if isinstance(mod, Dropout):
                k += 1
                mod.salt = k

    # ---- fused engine ------------------------------------------------------------------------
    def use_fused_step(self, **kw):
        """Route train_step through the fused HIP engine (engine.FusedTrainer)."""
        from .engine import FusedTrainer
        self._engine = FusedTrainer(self, **kw)
        return self._engine

    def use_sparse_step(self, **kw):
        """Route train_step through sparse_trainer.SparseTrainer: this model's own forward under autograd, table
        rows updated only where the batch touches them (any model kind)."""
        from .sparse_trainer import SparseTrainer
        self._engine = SparseTrainer(self, **kw)
        return self._engine

    def _check_indices(self):
        """The reference's nn.Embedding raises IndexError on an out-of-range index at once; the kernels here read the
        padding row instead and set a device flag, which is read once per epoch (and every few hundred steps by the
        trainers)."""
        eng = self._engine
        if eng is not None and hasattr(eng, "check"):
            eng.check()
        for mod in self.modules():
            for name in ("_tab", "_tab1"):
                holder = getattr(mod, name, None)
                if holder is not None and getattr(holder, "layout", None) is not None:
                    holder.layout.check_indices()

    def _flush_engine(self):
        """A touched-rows Adam trainer replays the skipped steps of a row only when the row is next read; before the
        tables are read or written as a whole every row is brought up to date."""
        eng = self._engine
        if eng is not None and hasattr(eng, "flush"):
            eng.flush()
        for mod in self.modules():              # trainers built directly on the model leave their stamps on the holders
            for name in ("_tab", "_tab1"):
                holder = getattr(mod, name, None)
                if holder is not None and hasattr(holder, "flush_rows"):
                    holder.flush_rows()

    def state_dict(self, *args, **kwargs):
        self._flush_engine()
        return super().state_dict(*args, **kwargs)

    def load_state_dict(self, *args, **kwargs):
        self._flush_engine()
        return super().load_state_dict(*args, **kwargs)

    # ---- loops (BaseModel.py:98-173) ---------------------------------------------------------
    def fit(self):
        loader = self.dataRecorder.train_loader
        if self.record:
            self.logger.info(f"model:{self.model_name}, batch size: {self.dataRecorder.batch_size}, "
                             f"embedding dim:{self.dataRecorder.embedding_dim}, optimizer: {self.optimizer_type}")
            self.logger.info(f"start training: {len(loader)} batches/epoch")
        self.best_auc = float("inf") if not self.fix_earlystop else -float("inf")
        for epoch in range(self.num_epochs):
            self.epoch_index = epoch
            self.train_epoch()
            if self.stop_training:
                break
            self.logger.info(f"=========== Epoch = {epoch + 1} end ===========")
        path = os.path.join(self.checkpoint, f"{self.model_name}.model")
        if os.path.exists(path):
            self.load_weights(path)

    def load_weights(self, checkpoint):
        self.to(self.device)
        state_dict = torch.load(checkpoint, map_location="cpu", weights_only=True)
        self.load_state_dict(state_dict)

    def train_epoch(self):
        self.train()
        train_loss = 0.0
        loader = self.dataRecorder.train_loader
        for inputs, labels in loader:
            loss = self.train_step(inputs, labels)
            train_loss += loss.item()
            if self.stop_training:
                break
        self._check_indices()
        self.last_train_loss = train_loss / max(len(loader), 1)
        self.logger.info(f"Train Loss: {self.last_train_loss:.6f}")
        self.eval_step()

    def train_step(self, inputs, labels):
        """BaseModel.py:141-150."""
        if self._engine is not None:
            return self._engine.step(inputs, labels)
        self.optimizer.zero_grad()
        labels = self.labels_to_device(labels)
        outputs = self.forward(inputs)
        loss = self.criterion(outputs.squeeze(1), labels.float())
        loss += self.regularization_loss()
        loss.backward()
        nn.utils.clip_grad_norm_(self.parameters(), max_norm=10)
        self.optimizer.step()
        return loss

    def eval_step(self):
        val_auc = self.evaluate(self.dataRecorder.valid_loader)
        self.checkpoint_and_earlystop(val_auc)
        self.train()

    def evaluate(self, data_loader):
        self.eval()
        self._flush_engine()
        with torch.no_grad():
            val_labels, val_preds = [], []
            for inputs, labels in data_loader:
                labels = self.labels_to_device(labels)
                outputs = self.forward(inputs)
                val_labels.append(labels.reshape(-1).cpu())
                val_preds.append(outputs.data.reshape(-1).cpu())
            self._check_indices()          # the predictions are on the host already: reading the flag adds no wait
            y = torch.cat(val_labels)
            p = torch.cat(val_preds)
            self.last_logloss = log_loss(y, p, eps=1e-7)
            val_auc = roc_auc(y, p)
            self.logger.info(f"AUC: {val_auc:.6f} - Logloss: {self.last_logloss:.6f}")
            return val_auc

    def checkpoint_and_earlystop(self, val_auc, delta=1e-6):
        """BaseModel.py:175-189."""
        improved = (val_auc > self.best_auc + delta) if self.fix_earlystop else (val_auc < self.best_auc - delta)
        if improved:
            self.best_auc = val_auc
            self.counter = 0
            os.makedirs(self.checkpoint, exist_ok=True)
            self.save_weights()
        else:
            self.counter += 1
            if self.reduce_lr_on_plateau:
                current_lr = self.lr_decay()
                self.logger.info("learning rate reduced to {:.6f}".format(current_lr))
            if self.counter >= self.patience:
                self.stop_training = True
                self.logger.info(f"======== Early stop at Epoch = {self.epoch_index} =========")

    def lr_decay(self, factor=0.1, min_lr=1e-6):
        reduced_lr = None
        if self.optimizer is not None:
            for param_group in self.optimizer.param_groups:
                reduced_lr = max(param_group["lr"] * factor, min_lr)
                param_group["lr"] = reduced_lr
        if self._engine is not None:
            reduced_lr = self._engine.decay_lr(factor, min_lr)
        return reduced_lr

    def save_weights(self):
        path = os.path.join(self.checkpoint, f"{self.model_name}.model")
        torch.save(self.state_dict(), path)
        self.logger.info(f"Best Model with AUC = {self.best_auc:.6f} saved to {path}")

    def features_to_divice(self, inputs):
        return {feature: t.to(self.device) for feature, t in inputs.items()}

    def labels_to_device(self, labels):
        return next(iter(labels.values())).to(self.device)

    def regularization_loss(self):
        """BaseModel.py:214-231: (lambda/2)*||p||^2 per weight/bias; nn.Embedding uses emb_regular."""
        reg_term = 0
        if self.net_regular or self.emb_regular:
            for _, module in self.named_modules():
                for p_name, param in module.named_parameters():
                    if param.requires_grad and p_name in ["weight", "bias"]:
                        if type(module) == nn.Embedding:
                            if self.emb_regular:
                                reg_term += (self.emb_regular / 2) * torch.norm(param, 2) ** 2
                        elif self.net_regular:
                            reg_term += (self.net_regular / 2) * torch.norm(param, 2) ** 2
        return reg_term
