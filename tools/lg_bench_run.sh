cd $GRAFT_REPO_ROOT
timeout -k 5 60 ./tools/lg_bench_0 8 1024 1
