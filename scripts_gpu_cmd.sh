cd /root/repo
python bench.py --steps 2 --warmup 1 2>&1 | tail -1 | tee gpurun_out/bench_line.json
export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/prof_r01 -o bench -- python /root/repo/bench.py --steps 1 --warmup 0 --no-cpu-baseline > /root/repo/gpurun_out/prof_bench.log 2>&1
tail -1 /root/repo/gpurun_out/prof_bench.log | cut -c1-300
