"""Meta-test path: metric records (CPU, known answers) and, on the GPU, the batched fit+predict against the
reference-shaped per-task loop and the float64 oracle."""
import numpy as np
import pytest
import torch

from adkf_ift_amd import evaluate as E


def test_binary_metrics_known_answers():
    labels = np.array([1, 0, 1, 1, 0, 0, 1, 0])
    preds = np.array([0.9, 0.2, 0.6, 0.4, 0.55, 0.1, 0.8, 0.3])
    m = E.compute_binary_task_metrics(preds, labels)
    # hard predictions: 1 0 1 0 1 0 1 0 -> TP 3, FP 1, FN 1, TN 3
    assert m.size == 8 and m.acc == pytest.approx(0.75) and m.balanced_acc == pytest.approx(0.75)
    assert m.prec == pytest.approx(0.75) and m.recall == pytest.approx(0.75) and m.f1 == pytest.approx(0.75)
    assert m.kappa == pytest.approx(0.5)
    # ranking: positives {.9,.8,.6,.4}, negatives {.55,.3,.2,.1}: 15 of 16 pairs ordered correctly
    assert m.roc_auc == pytest.approx(15 / 16)
    # AP = mean over positives of precision at their rank: ranks 1,2,3,5 -> (1 + 1 + 1 + 4/5) / 4
    assert m.avg_precision == pytest.approx((1 + 1 + 1 + 0.8) / 4)


def test_binary_metrics_single_class_and_zero_division():
    m = E.compute_binary_task_metrics([0.1, 0.2, 0.3], [0, 0, 0])
    assert m.roc_auc == 0.0                       # fs_mol/utils/metrics.py:46-49
    assert m.prec == 1.0 and m.recall == 1.0 and m.f1 == 1.0   # zero_division=1


def test_numeric_metrics_out_of_sample_r2():
    y = np.array([1.0, -1.0, 2.0, 0.5])
    p = np.array([0.5, -0.5, 1.0, 0.0])
    m = E.compute_numeric_task_metrics(p, y)
    assert m.mse == pytest.approx(np.mean((y - p) ** 2)) and m.mae == pytest.approx(np.mean(np.abs(y - p)))
    assert m.r2 == pytest.approx(1 - ((y - p) ** 2).sum() / (y ** 2).sum())   # baseline = training mean = 0, not mean(y)


def test_aggregation_over_samples_and_tasks():
    a = [E.NumericEvalMetrics(4, 1.0, 0.5, 0.2), E.NumericEvalMetrics(4, 3.0, 1.5, 0.4)]
    b = [E.NumericEvalMetrics(4, 2.0, 1.0, 0.0)]
    per = E.avg_task_metrics_list(a)
    assert per["mse"] == (2.0, 1.0) and per["r2"][0] == pytest.approx(0.3)
    allm = E.avg_metrics_over_tasks({"a": a, "b": b})
    assert allm["mse"][0] == pytest.approx(2.0) and allm["mse"][1] == pytest.approx(0.0)
    assert allm["r2"][0] == pytest.approx(0.15)


@pytest.mark.gpu
@pytest.mark.parametrize("numeric", [False, True])
def test_batched_meta_test_equals_reference_shaped_loop(numeric):
    from adkf_ift_amd.meta_batch import collate_meta_batch, meta_features
    from adkf_ift_amd.models import ADKTModel
    from oracle import gp_oracle as O
    from test_meta_batch import random_task, small_model

    dev = torch.device("cuda:0")
    torch.manual_seed(1)
    model = ADKTModel(small_model(numeric)).to(dev)
    with torch.no_grad():
        for blk in model.graph_feature_extractor.gnn.gnn_blocks:
            blk.alpha.fill_(0.5)
    tasks = [random_task(16, 40, 21).to(dev), random_task(13, 9, 22).to(dev), random_task(16, 130, 23).to(dev)]
    got = E.evaluate_tasks(model, tasks, names=["a", "b", "c"], tasks_per_call=2)
    mb = collate_meta_batch(tasks).to(dev)
    preds, var, phi, _ = E.meta_test(model, mb, want_var=True)
    with torch.no_grad():
        Z_s, Z_q = meta_features(model.eval(), mb)
    y_s, _ = mb.labels(numeric)
    for k, (name, task) in enumerate(zip("abc", tasks)):
        lab = task.query_numeric_labels if numeric else task.query_labels
        want = E.run_on_batches(model, [task], [task.query_labels], [task.query_numeric_labels], train=False)
        for f in ("size", "acc", "avg_precision") if not numeric else ("size", "mse", "r2"):
            assert getattr(got[name], f) == pytest.approx(getattr(want, f), rel=2e-3, abs=2e-3), (name, f)
        # the GP tail against the float64 oracle at the fitted hyper-parameters
        ns, nq = task.num_support_samples, task.num_query_samples
        mean, v = O.predict(Z_s[k, :ns].double().cpu(), y_s[k, :ns].double().cpu(), Z_q[k, :nq].double().cpu(),
                            phi[k].double().cpu(), 1)[:2]
        raw = preds[k, :nq].double().cpu() if numeric else torch.logit(preds[k, :nq].double().cpu())
        assert (raw - mean).abs().max().item() <= 1e-4 * max(1.0, mean.abs().max().item())
        vv = v if v.dim() == 1 else v.diagonal()
        assert (var[k, :nq].double().cpu() - vv).abs().max().item() <= 1e-4 * vv.abs().max().item()
        assert float(preds[k, nq:].abs().max() if nq < preds.shape[1] else 0.0) == 0.0
    assert E.run_on_batches(model, [tasks[0]], [tasks[0].query_labels], [tasks[0].query_numeric_labels], train=True) is None
