set -x
cd tools && hipcc -O3 --offload-arch=gfx950 -std=c++17 -I ../adkf_ift_amd/csrc sweepw_bench.hip -o sweepw_bench 2>/dev/null; cd ..
timeout -k 10 120 tools/sweepw_bench 256 > gpurun_out/sweepw.log 2>&1; echo "sweepw rc $?"
cat gpurun_out/sweepw.log
