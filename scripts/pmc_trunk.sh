# PMC counters for the dominant kernels (separate pass from timing runs; see MI355X_MICROARCH.md "rocprofv3 PMC slots")
export TMPDIR=/tmp; OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc; rm -rf $OUT; mkdir -p $OUT; cd /tmp
rocprofv3 -L > $OUT/counters_list.txt 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $OUT/p1 -o k -- python3 $GRAFT_REPO_ROOT/scripts/kbench.py trunk > $OUT/p1.log 2>&1; echo "pmc1 exit=$?"
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_COEXEC_CYCLES --kernel-trace --output-format csv -d $OUT/p2 -o k -- python3 $GRAFT_REPO_ROOT/scripts/kbench.py trunk > $OUT/p2.log 2>&1; echo "pmc2 exit=$?"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/p3 -o k -- python3 $GRAFT_REPO_ROOT/scripts/kbench.py trunk > $OUT/p3.log 2>&1; echo "pmc3 exit=$?"
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/p4 -o k -- python3 $GRAFT_REPO_ROOT/scripts/kbench.py trunk > $OUT/p4.log 2>&1; echo "pmc4 exit=$?"
ls -R $OUT | head -30
