"""Meta-test path: adapt the GP to each test task's support set, predict its query molecules, score.

Mirrors ``run_on_batches(train=False)`` / ``evaluate_adkt_model`` (fs_mol/utils/adaptive_dkt_utils.py:70-175) and the
metric records of fs_mol/utils/metrics.py:21-60,110-150 (same field names, same conventions: predictions >= 0.5 count as
positive, ``zero_division=1``, ROC-AUC 0.0 for single-class tasks, out-of-sample R^2 against a zero baseline because
regression labels are standardised with support statistics).

Two entry points:

* ``run_on_batches``      - the reference-shaped loop, one query batch at a time through ``ADKTModel.forward``;
* ``meta_test``           - every task of a ``MetaBatch`` at once: ONE feature-extractor forward (no grad), then
                            ``adkf_init_params -> adkf_fit -> adkf_predict`` on the whole batch (the fit's A^-1, alpha
                            are reused by the prediction).  No hypergradient is involved.
* ``adapt_and_test``      - MoleculeNet's test protocol (MoleculeNet/chem_lib/models/adkfift_trainer.py:222-283): per test task,
                            ``update_step_test`` hypergradient steps on the task's adaptation batches from the saved weights,
                            then the final fit on the support set and the prediction of the query loader.
"""
from __future__ import annotations

import dataclasses
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from .meta_batch import DKTBatch, MetaBatch, collate_meta_batch, meta_features


@dataclass(frozen=True)
class BinaryEvalMetrics:
    size: int
    acc: float
    balanced_acc: float
    f1: float
    prec: float
    recall: float
    roc_auc: float
    avg_precision: float
    kappa: float


@dataclass(frozen=True)
class NumericEvalMetrics:
    size: int
    mse: float
    mae: float
    r2: float


def compute_binary_task_metrics(predictions, labels) -> BinaryEvalMetrics:
    from sklearn import metrics as skm

    predictions = np.asarray(predictions, dtype=np.float64)
    labels = np.asarray(labels).astype(bool)
    hard = predictions >= 0.5
    single_class = labels.all() or not labels.any()
    return BinaryEvalMetrics(
        size=len(predictions),
        acc=float(skm.accuracy_score(labels, hard)),
        balanced_acc=float(skm.balanced_accuracy_score(labels, hard)),
        f1=float(skm.f1_score(labels, hard, zero_division=1)),
        prec=float(skm.precision_score(labels, hard, zero_division=1)),
        recall=float(skm.recall_score(labels, hard, zero_division=1)),
        roc_auc=0.0 if single_class else float(skm.roc_auc_score(labels, predictions)),
        avg_precision=float(skm.average_precision_score(labels, predictions)),
        kappa=float(skm.cohen_kappa_score(labels, hard)),
    )


def r2_score_os(y_true, y_pred, y_train_mean: float = 0.0) -> float:
    """Out-of-sample R^2: the baseline is the TRAINING mean (0 after standardisation), fs_mol/utils/metrics.py:127-143."""
    y_true = np.asarray(y_true, dtype=np.float64)
    y_pred = np.asarray(y_pred, dtype=np.float64)
    assert y_true.shape == y_pred.shape
    den = ((y_true - y_train_mean) ** 2).sum()
    assert den != 0
    return float(1.0 - ((y_true - y_pred) ** 2).sum() / den)


def compute_numeric_task_metrics(predictions, labels) -> NumericEvalMetrics:
    p = np.asarray(predictions, dtype=np.float64)
    y = np.asarray(labels, dtype=np.float64)
    assert p.shape == y.shape
    return NumericEvalMetrics(size=len(p), mse=float(((p - y) ** 2).mean()), mae=float(np.abs(p - y).mean()), r2=r2_score_os(y, p))


def avg_task_metrics_list(results: Sequence) -> Dict[str, Tuple[float, float]]:
    """mean / std of every field over repeated samples of one task (metrics.py:83-93, 176-186)."""
    out = {}
    for f in dataclasses.fields(type(results[0])):
        v = [getattr(r, f.name) for r in results]
        out[f.name] = (float(np.mean(v)), float(np.std(v)))
    return out


def avg_metrics_over_tasks(task_results: Dict[str, Sequence]) -> Dict[str, Tuple[float, float]]:
    """mean / std across tasks of the per-task means (metrics.py:63-80, 153-173)."""
    per_task = {k: avg_task_metrics_list(v) for k, v in task_results.items()}
    first = next(iter(task_results.values()))[0]
    out = {}
    for f in dataclasses.fields(type(first)):
        v = [m[f.name][0] for m in per_task.values()]
        out[f.name] = (float(np.mean(v)), float(np.std(v)))
    return out


def _score(model, preds: np.ndarray, labels: np.ndarray):
    if model.config.use_numeric_labels:
        return compute_numeric_task_metrics(preds, labels)
    return compute_binary_task_metrics(preds, labels)


def run_on_batches(model, batches: List[DKTBatch], batch_labels: List[torch.Tensor], batch_numeric_labels: List[torch.Tensor],
                   train: bool = False):
    """Reference-shaped: per query batch {train-mode forward (re-initialises the GP on the support set), inner fit,
    eval-mode forward}; predictions are ``sigmoid(mean)`` (classification) or ``mean`` (regression)."""
    from .models import fit_gpytorch_scipy

    if train:
        assert len(batches) == 1
    preds, labels = [], []
    for feats, lab, num in zip(batches, batch_labels, batch_numeric_labels):
        model.train()
        _ = model(feats, train_loss=True)
        fit_gpytorch_scipy(model.mll)
        if not train:
            model.eval()
            with torch.no_grad():
                post = model(feats, train_loss=None)
                if model.config.use_numeric_labels:
                    preds.append(post.mean.detach().cpu().numpy())
                    labels.append(num.detach().cpu().numpy())
                else:
                    preds.append(torch.sigmoid(post.mean).detach().cpu().numpy())
                    labels.append(lab.detach().cpu().numpy())
    if train:
        return None
    return _score(model, np.concatenate(preds), np.concatenate(labels))


@torch.no_grad()
def meta_test(model, mb: MetaBatch, max_evals: int = 200, gtol: float = 1e-5, ftol: float = 2.22e-9, want_var: bool = False):
    """All tasks at once.  Returns (predictions [T, Nq_max], variance or None, phi* [T, 3], n_evals [T]); padded query
    slots hold 0.  Classification predictions are already passed through the sigmoid."""
    from . import gp_ops

    cfg = model.config
    was_training = model.training
    model.eval()
    Z_s, Z_q = meta_features(model, mb)
    y_s, _ = mb.labels(cfg.use_numeric_labels)
    dev = Z_s.device
    priors = torch.empty(mb.num_tasks, 4, dtype=torch.float32, device=dev)
    b = gp_ops.GPBatch(Z_s.float().contiguous(), y_s.to(dev).float().contiguous(), priors, cfg.gp_kernel,
                       Z_q=Z_q.float().contiguous(), n_s=mb.n_s, n_q=mb.n_q, ard=cfg.use_ard)
    phi0, _ = gp_ops.init_params_batch(b, cfg.use_numeric_labels, cfg.use_lengthscale_prior)
    b.flags = gp_ops.REUSE_DIST | gp_ops.DEFER_REFINE     # (the prediction redoes ill-conditioned tasks in float64 itself)
    phi, _, _, n_evals, info = gp_ops.fit(b, phi0, max_evals, gtol, ftol)
    gp_ops.check_info(info, "meta-test inner fit")
    b.flags = gp_ops.REUSE_DIST | gp_ops.REUSE_INNER
    mean, var, _, info = gp_ops.predict(b, phi, want_var=want_var)
    gp_ops.check_info(info, "meta-test prediction")
    if was_training:
        model.train()
    preds = mean if cfg.use_numeric_labels else torch.sigmoid(mean) * mb.q_mask.to(dev)
    return preds, var, phi, n_evals


def evaluate_tasks(model, tasks: Sequence[DKTBatch], names: Optional[Sequence[str]] = None, tasks_per_call: int = 64,
                   max_evals: int = 200) -> Dict[str, object]:
    """``evaluate_adkt_model`` for tasks that are already in memory: per-task metric records, ``tasks_per_call`` tasks
    per library call (one disconnected graph each)."""
    names = list(names) if names is not None else [f"task{i}" for i in range(len(tasks))]
    dev = model.device
    out: Dict[str, object] = {}
    for lo in range(0, len(tasks), tasks_per_call):
        chunk = tasks[lo:lo + tasks_per_call]
        mb = collate_meta_batch(chunk).to(dev)
        preds, _, _, _ = meta_test(model, mb, max_evals=max_evals)
        preds = preds.cpu().numpy()
        for k, task in enumerate(chunk):
            nq = task.num_query_samples
            lab = task.query_numeric_labels if model.config.use_numeric_labels else task.query_labels
            out[names[lo + k]] = _score(model, preds[k, :nq], lab.detach().cpu().numpy())
    return out


def adapt_and_test(model, optimizer, saved_state_dict, adapt_data, eval_data, update_step_test: int = 1, clip_value: float = 1.0):
    """One test task of ``Meta_Trainer.test_step`` (MoleculeNet/chem_lib/models/adkfift_trainer.py:225-283) for an ``ADKFModel``.

    ``adapt_data`` / ``eval_data``: dicts with ``s_data``, ``s_label`` and ``data_loader`` (an iterable of query batches with
    ``.to(device)`` and ``.y``), as ``get_data_sample(task_id, train=False)`` builds them.  From ``saved_state_dict`` (:226), for the
    first ``update_step_test`` batches of the adaptation loader (:229-271): re-initialise and fit the GP tail on the support set,
    IFT hypergradient of the predictive loss of the batch (fused HIP path: ONE extractor backward), clip-by-global-norm
    ``clip_value``, ``optimizer.step()``.  Then (:273-281) fit on the evaluation support set and return
    ``(sigmoid(posterior mean), labels)`` over the evaluation loader, plus the per-step outer losses."""
    from .hypergradient import cauchy_hypergradient
    from .models import fit_gpytorch_scipy

    model.load_state_dict(saved_state_dict)
    dev = model.device
    losses = []
    if update_step_test > 0:
        for i, batch in enumerate(adapt_data["data_loader"]):
            optimizer.zero_grad()
            batch = batch.to(dev) or batch          # torch_geometric's Batch.to returns the moved batch; stand-ins may move in place
            model.train()
            model(s_data=adapt_data["s_data"], q_data=None, s_label=adapt_data["s_label"], train_loss=True)      # a3/a4
            fit_gpytorch_scipy(model.mll)                                                                         # a7
            model.train()
            f_outer, f_inner = model.task_losses((adapt_data["s_data"], batch, adapt_data["s_label"]))
            loss = cauchy_hypergradient(f_outer, f_inner, tuple(model.feature_extractor_params()), tuple(model.gp_params()), dev,
                                        ignore_grad_correction=False)                                             # a9
            torch.nn.utils.clip_grad_norm_(model.feature_extractor_params(), clip_value)
            optimizer.step()
            losses.append(float(loss))
            if i >= update_step_test - 1:
                break
    model.train()
    model(s_data=eval_data["s_data"], q_data=None, s_label=eval_data["s_label"], train_loss=True)
    fit_gpytorch_scipy(model.mll)
    model.eval()
    with torch.no_grad():
        q_preds, q_labels = model.forward_query_loader(eval_data["s_data"], eval_data["data_loader"], train_loss=None,
                                                       s_label=eval_data["s_label"])
    return q_preds, q_labels, losses
