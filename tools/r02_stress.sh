set -x
python -m pytest tests/test_gpu_stress.py -x -q -s -k "random_shapes or ill_conditioned" > gpurun_out/t_stress_default.log 2>&1; echo "default rc $?"
ADKF_LDL_THRESHOLD=0 python -m pytest tests/test_gpu_stress.py -x -q -s -k "random_shapes or ill_conditioned" > gpurun_out/t_stress_ldl0.log 2>&1; echo "ldl0 rc $?"
python -m pytest tests/test_gpu_gnn.py -x -q -s > gpurun_out/t_gnn.log 2>&1; echo "gnn rc $?"
