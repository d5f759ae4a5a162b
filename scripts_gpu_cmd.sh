cd /root/repo
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py tests/test_gpu_adm.py -x -q -m gpu 2>&1 | tail -5
python tools/layer_report.py > gpurun_out/layer_report.txt 2>&1; head -22 gpurun_out/layer_report.txt
