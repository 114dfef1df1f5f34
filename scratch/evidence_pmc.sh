#!/bin/bash
# PMC passes of the round's evidence (own rocprofv3 runs, counters only)
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/ev
cd $R
bash scratch/pmc_bench.sh ev_pmc_rw "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" conv_ > gpurun_out/ev/pmc_rw.txt 2>&1
echo "pmc rw done"
bash scratch/pmc_any.sh ev_pmc_mfma "SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" conv_ mb_one.py 256 256 3 1 32 fwd
echo "pmc mfma done"
bash scratch/pmc_any.sh ev_pmc_corr_sq "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS GRBM_GUI_ACTIVE" correlation mb_corr.py
bash scratch/pmc_any.sh ev_pmc_corr_rd "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_HIT_sum TCC_MISS_sum" correlation mb_corr.py
echo "pmc corr done"
