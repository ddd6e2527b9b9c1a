set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof_cfg
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
CMD="python $R/tools/config_bench.py C4 C5 h256 512"
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/pmc$i -- $CMD > $O/pmc$i.log 2>&1
  python $R/tools/pmc_summary.py $O/pmc$i $O/pmc_configs_$i.json
done
rm -rf $O/pmc[0-9]
ls -la $O
