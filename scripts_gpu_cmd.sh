cd /root/repo
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py tests/test_gpu_adm.py -x -q -m gpu 2>&1 | tail -5
python bench.py --steps 2 --warmup 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['achieved'], d['roofline']['whole_loop_tflops'])"
python tools/adm_probe.py 2>&1 | grep "fp16:"
