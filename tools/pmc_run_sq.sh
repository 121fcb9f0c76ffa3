cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT --kernel-include-regex "k_matvec_pairs_fast|k_assemble_fast|k_cg_update|k_matvec_finish" --output-format csv -d $R/gpurun_out/pmc_r01e_a -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmc_r01e_a.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_SMEM SQ_WAIT_INST_LDS --kernel-include-regex "k_matvec_pairs_fast|k_assemble_fast|k_cg_update|k_matvec_finish" --output-format csv -d $R/gpurun_out/pmc_r01e_b -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmc_r01e_b.log 2>&1
ls $R/gpurun_out/pmc_r01e_a/* $R/gpurun_out/pmc_r01e_b/* | head
