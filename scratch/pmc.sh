cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pmc
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d gpurun_out/pmc/sq -- python3 bench.py --steps 6 --warmup 12 --no-cpu-baseline > gpurun_out/pmc/sq.json 2> gpurun_out/pmc/sq.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc/fetch -- python3 bench.py --steps 6 --warmup 12 --no-cpu-baseline > gpurun_out/pmc/fetch.json 2> gpurun_out/pmc/fetch.err
rocprofv3 --pmc WRITE_SIZE SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmc/write -- python3 bench.py --steps 6 --warmup 12 --no-cpu-baseline > gpurun_out/pmc/write.json 2> gpurun_out/pmc/write.err
ls -R gpurun_out/pmc | head -30; tail -3 gpurun_out/pmc/sq.err
