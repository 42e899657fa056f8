#!/bin/bash
# PMC passes (separate runs, no trace domains) for the kernels whose bound DESIGN.md states from instruction counts rather than
# from bytes: the MI recompute passes, the tolerance-mode lean LK kernel and the grid loop.  Outputs: gpurun_out/r02pmc/<wl>/summary.txt
cd "$GRAFT_REPO_ROOT" || exit 1
G1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU"
G2="SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE"
G3="SQ_INSTS_VALU_MFMA_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_MISC"
G4="FETCH_SIZE"
G5="WRITE_SIZE"
timeout 500 bash tools/pmc_collect.sh gpurun_out/r02pmc/mi "--workload mi --steps 3 --warmup 2" "$G1" "$G2" "$G3" "$G4" "$G5" > /dev/null 2>&1
timeout 300 bash tools/pmc_collect.sh gpurun_out/r02pmc/lean "--mode lean --steps 10 --warmup 2 --no-lean" "$G1" "$G2" "$G4" "$G5" > /dev/null 2>&1
timeout 300 bash tools/pmc_collect.sh gpurun_out/r02pmc/grid "--workload grid --steps 10 --warmup 2" "$G1" "$G2" > /dev/null 2>&1
for w in mi lean grid; do echo "== $w"; cat gpurun_out/r02pmc/$w/summary.txt | head -80; done
