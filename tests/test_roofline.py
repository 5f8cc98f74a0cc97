"""CPU: the algorithmic-work model bench.py prices the kernels with is the one SURVEY.md section 8(d) freezes."""
from adkf_ift_amd import roofline


def test_flops_per_task_match_the_survey_figures():
    c2 = roofline.flops_per_task(128, 128, 256, 20)
    assert abs(c2["total"] / 1e6 - 163.6) < 0.1                   # "C2: ~163.6 MFLOP/task"
    assert abs(c2["inner_fit"] / 1e6 - 41.9) < 0.1                # I * N^3
    assert abs(roofline.flops_per_task(32, 32, 64, 20)["total"] / 1e6 - 2.56) < 0.01      # C1
    assert abs(roofline.flops_per_task(1024, 1024, 512, 20)["total"] / 1e9 - 54.76) < 0.01  # C5
    # the pieces add up and scale as stated: distance builds 2d(N^2 + N Nq + Nq^2), dZ GEMMs twice that
    assert abs(sum(v for k, v in c2.items() if k != "total") - c2["total"]) < 1.0
    assert roofline.PEAK_FP32_TFLOPS == 157.3
