cd /root/repo
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py tests/test_gpu_adm.py -m gpu -q 2>&1 | tail -6
python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('c2', d['value'], d['ms_per_step'])"
python tools/adm_probe.py 4 2>&1 | grep -E "^fp|halo_f16|gather"
