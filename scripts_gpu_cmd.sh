export TMPDIR=/tmp; cd /tmp
O=/root/repo/gpurun_out
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_mfma -o p -- python /root/repo/tools/forward_once.py 2 > $O/pmc1.log 2>&1; tail -2 $O/pmc1.log
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o p -- python /root/repo/tools/forward_once.py 2 > $O/pmc2.log 2>&1; tail -1 $O/pmc2.log
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o p -- python /root/repo/tools/forward_once.py 2 > $O/pmc3.log 2>&1; tail -1 $O/pmc3.log
ls -la $O/pmc_mfma $O/pmc_fetch | head -20
