"""CPU: the synthetic workload of SURVEY.md section 8(d) is the same on every rank and every box."""
import torch

from adkf_ift_amd.synthetic import make_outer_weight, make_tasks


def test_tasks_are_deterministic_and_shard_consistently():
    whole = make_tasks(6, 8, 5, N_q=7)
    again = make_tasks(6, 8, 5, N_q=7)
    for a, b in zip((whole.X_s, whole.X_q, whole.y_s, whole.y_q, whole.W), (again.X_s, again.X_q, again.y_s, again.y_q, again.W)):
        assert torch.equal(a, b)
    # rank r of a weak-scaling run builds tasks [r T, (r + 1) T): exactly the slice of the larger batch
    part = make_tasks(2, 8, 5, N_q=7, first_task=4)
    assert torch.equal(part.X_s, whole.X_s[4:6]) and torch.equal(part.y_q, whole.y_q[4:6])
    assert torch.equal(part.W, make_outer_weight(5))
    assert whole.X_s.dtype == torch.float32 and whole.y_s.dtype == torch.float32


def test_labels_follow_the_survey_recipe():
    cls = make_tasks(3, 16, 4)
    assert set(cls.y_s.unique().tolist()) <= {-1.0, 1.0} and set(cls.y_q.unique().tolist()) <= {-1.0, 1.0}
    reg = make_tasks(3, 16, 4, regression=True)
    # standardised with the SUPPORT statistics (fs_mol/data/dkt.py:91-97): zero mean, unit (unbiased) std on the support set
    assert float(reg.y_s.mean(1).abs().max()) < 1e-5
    assert float((reg.y_s.std(1) - 1).abs().max()) < 1e-5
    Zs, Zq = cls.features()
    assert Zs.shape == (3, 16, 4) and Zq.shape == (3, 16, 4)
