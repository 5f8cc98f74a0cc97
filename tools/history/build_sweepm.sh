#!/bin/bash
# builds tools/sweepm_bench variants: ./tools/build_sweepm.sh <suffix> <extra -D flags...>
set -e
suffix=$1; shift
cd /tmp && hipcc -O3 --offload-arch=gfx950 -std=c++17 -w "$@" -I /root/repo/adkf_ift_amd/csrc /root/repo/tools/sweepm_bench.hip -o /root/repo/tools/sweepm_bench$suffix
