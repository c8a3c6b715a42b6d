set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r03
cd /tmp
rocprofv3 -L > $GRAFT_REPO_ROOT/gpurun_out/r03/counters_list.txt 2>&1
for pass in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" "SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM" "SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_WAIT_INST_VMEM SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_VALU GRBM_GUI_ACTIVE SQ_LDS_DATA_FIFO_FULL"; do
  tag=$(echo $pass | cut -d' ' -f1)
  timeout 300 rocprofv3 --kernel-trace --pmc $pass -d $GRAFT_REPO_ROOT/gpurun_out/r03/pmc_$tag --output-format csv -- python $GRAFT_REPO_ROOT/tools/gemm_one.py 65536 10240 1280 25 4 > $GRAFT_REPO_ROOT/gpurun_out/r03/pmc_$tag.log 2>&1
done
cd $GRAFT_REPO_ROOT
python - <<'P'
import csv, glob, collections
for d in sorted(glob.glob("gpurun_out/r03/pmc_SQ_*") + glob.glob("gpurun_out/r03/pmc_GRBM*")):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "gemm_kernel_v7" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in acc.items():
            print(f"{k:34s} n={len(v)} last={v[-1]:.4g} mean={sum(v)/len(v):.4g}")
P
( timeout 1500 python -m pytest tests/test_pipeline_gpu.py tests/test_fullsize_properties_gpu.py tests/test_compat_gpu.py tests/test_mx8_gpu.py tests/test_config4_gpu.py -q -s -x 2>&1 | tail -40 ) > gpurun_out/r03/tests5.log 2>&1
tail -40 gpurun_out/r03/tests5.log
