python -m pytest tests -m gpu -x -q -s > gpurun_out/t_all.log 2>&1; echo "all rc $?"
grep -a "passed\|failed\|Error\|error" gpurun_out/t_all.log | tail -8
