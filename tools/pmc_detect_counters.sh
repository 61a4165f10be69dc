cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for set in "SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_LEVEL_WAVES GRBM_GUI_ACTIVE"; do
rm -rf /tmp/pd; LANES=64 FLAGS=1 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pd -- python $R/tools/detect_bench.py > /dev/null 2>&1
python $R/tools/pmc_summary.py $(find /tmp/pd -name "*counter_collection.csv" | head -1) | grep -E "kernel|k_describe|k_fast|k_resize|k_select"
done
