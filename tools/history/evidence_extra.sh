# extra round-2 evidence (run through gpurun): GEMM ablations, per-task kernel phases, evaluation phases of k_inner, C3 kernel breakdown
cd $GRAFT_REPO_ROOT
{ echo "# tools/gemm_bench.hip: distance GEMM at C2 (256 tasks x 10 tiles of 64x64, K = 256); ablate bits: 1 no fragment reads, 2 no barriers, 4 no operand loads after the first chunk, 8 no epilogue"; for a in 0 1 2 4 8 15; do timeout -k 5 60 tools/gemm_bench_$a | tail -2 | head -1; done; } > gpurun_out/r02_gemm_ablation.txt 2>&1
{ echo "# tools/small_bench.hip (-DADKF_STAMP_SMALL=1): s_memtime cycles of the phases of k_hess and k_outer_factor, 256 tasks x 128 points (single cold launch for k_outer_factor)"; timeout -k 5 100 tools/small_bench; } > gpurun_out/r02_small_kernel_phases.txt 2>&1
{ echo "# tools/eval_phases.py: s_memtime cycles of one search evaluation inside k_inner (C2 shape, workgroup 8, lane 0)"; python tools/eval_phases.py 0 2>&1 | tail -9; } > gpurun_out/r02_eval_phases.txt 2>&1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_c3 -o c3 -- python tools/bench_c3.py --steps 5 --warmup 2 > gpurun_out/prof_c3.log 2>&1
cat gpurun_out/r02_gemm_ablation.txt gpurun_out/r02_eval_phases.txt | tail -20
