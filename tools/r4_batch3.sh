#!/bin/bash
# round-4 GPU batch 3: memset-node probe; MFMA / LDS / wait counters of the matrix-core kernels (conv3x3, generator, value_proj, mixing, out_proj)
cd $GRAFT_REPO_ROOT
out=gpurun_out/r4b3; mkdir -p $out
timeout -k 10 120 python3 tools/graph_memset_probe.py $out/graph_memset_probe > $out/memset_probe.log 2>&1; echo "memset probe rc=$?"; head -c 5000 $out/memset_probe.log
tools/gpu_pmc_kernel.sh r4b3_pmc conv3x3_f16x3 \
  "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16" \
  "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_MFMA_MOPS_BF16" \
  "GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_VMEM_TA_ADDR_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_VALU_MFMA_COEXEC_CYCLES" > $out/pmc_conv.log 2>&1
python3 tools/pmc_all_kernels.py gpurun_out/r4b3_pmc $out/pmc_mfma_kernels.json conv3x3_f16x3 generator_ws gemm_split_kernel value_proj mixing_c64 conv3x3s2 sasa_mfma rowgemm > $out/pmc_all.log 2>&1; cat $out/pmc_all.log
LOWPREC_ONLY="i16,pyramid f16" timeout -k 10 500 python3 tools/exp_lowprec.py $out/exp_lowprec_block16.json > $out/exp_lowprec_block16.log 2>&1; echo "lowprec rc=$?"; grep -v amdgpu.ids $out/exp_lowprec_block16.log
