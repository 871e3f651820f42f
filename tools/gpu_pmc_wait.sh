#!/bin/bash
# What the fused kernel's waves wait on: four rocprofv3 --pmc passes (counters only) per kernel layout.
#   bash tools/gpu_pmc_wait.sh <tag> [params]     -> gpurun_out/<tag>_hp/summary.txt (k_hp) and <tag>_col/summary.txt (k_stream, variant 20)
TAG=${1:-wait}; PSET=${2:-D}
G1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY"
G2="SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INST_LEVEL_SMEM"
G3="SQ_IFETCH SQ_IFETCH_LEVEL SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL"
G4="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_THREAD_CYCLES_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU GRBM_GUI_ACTIVE"
KERN=k_hp bash $GRAFT_REPO_ROOT/tools/gpu_pmc_groups.sh ${TAG}_hp "--params $PSET" "$G1" "$G2" "$G3" "$G4" || exit 1
KERN=k_stream bash $GRAFT_REPO_ROOT/tools/gpu_pmc_groups.sh ${TAG}_col "--params $PSET --variant 20" "$G1" "$G2" "$G3" "$G4"
