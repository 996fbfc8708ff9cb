# PMC passes over the synthesis pulse kernel (run on the GPU box): bash scripts/syn_pmc.sh <tag>
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
rm -rf /tmp/spmc
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVES SQ_INSTS_SALU" \
           "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM SQ_INSTS_SMEM" \
           "SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_IFETCH" \
           "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_IFETCH_LEVEL SQ_INSTS_FLAT SQ_INSTS_BRANCH"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d /tmp/spmc/p$i -- python3 $R/scripts/syn_only.py 64 2 > /tmp/spmc_p$i.log 2>&1
done
python3 $R/scripts/pmc_summary.py /tmp/spmc syn_pulse | tee $O/$1_syn_pmc.txt
