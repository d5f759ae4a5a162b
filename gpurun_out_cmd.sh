cd /root/repo
python __graft_entry__.py smoke 2>&1 | tail -3
python bench.py --steps 1 --warmup 1 2>&1 | tail -5
export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/prof_r01 -o bench -- python /root/repo/bench.py --steps 1 --warmup 0 --no-cpu-baseline > /root/repo/gpurun_out/prof_bench.log 2>&1
tail -3 /root/repo/gpurun_out/prof_bench.log
ls -R /root/repo/gpurun_out/prof_r01 | head -20
