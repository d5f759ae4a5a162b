cd /root/repo
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -m gpu -q 2>&1 | tail -3
for i in 1 2; do
DDNM_NO_FUSED_GN=1 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('unfused', d['value'], d['ms_per_step'])"
python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fused  ', d['value'], d['ms_per_step'])"
done
