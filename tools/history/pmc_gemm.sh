# PMC passes over the default bench command for the batched-GEMM kernels (run through gpurun)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B="python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-parity --converge-steps 0"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmc_g1 -o p -- $B > gpurun_out/pmc_g1.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_WAVES --kernel-trace --output-format csv -d gpurun_out/pmc_g2 -o p -- $B > gpurun_out/pmc_g2.log 2>&1
python - <<'P'
import csv, collections
for d in ('pmc_g1','pmc_g2'):
    agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
    import glob
    f=glob.glob(f'gpurun_out/{d}/*counter_collection.csv')[0]
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'][:60]
        agg[k][r['Counter_Name']]+=float(r['Counter_Value'])
    for k,v in agg.items():
        if 'bgemm' in k or 'k_inner' in k or 'Cijk' in k:
            print(k, {a: f'{b:.3g}' for a,b in v.items()})
P
