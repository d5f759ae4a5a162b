cd /root/repo
for w in c3 c4 c5; do
  python bench.py --workload $w --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/bench_$w.json
  python -c "import json; d=json.load(open('gpurun_out/bench_$w.json')); print('$w', d['value'], d['ms_per_step'], d['config'].get('whole_loop_tflops_per_gpu'))"
done
