#!/bin/bash
set +e
cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out
for B in 32 64; do
  echo "== h16 B=$B"; B=$B timeout 300 python tools/cls_step.py 4 2>&1 | tail -1 | tee -a gpurun_out/c5_time.log
done
for G in 4 8; do
  echo "== c5 G=$G"; DDNM_CLS_GROUP=$G timeout 600 python bench.py --workload c5 --steps 2 --warmup 1 --no-cpu-baseline --no-roofline 2> gpurun_out/c5_g$G.err | tee gpurun_out/c5_g$G.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done
echo "== c5 gen1"; DDNM_CLS_GEN1=1 timeout 600 python bench.py --workload c5 --steps 2 --warmup 1 --no-cpu-baseline --no-roofline 2> gpurun_out/c5_gen1.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
timeout 900 python -m pytest tests/test_classifier.py tests/test_gpu_full_configs.py -q -m gpu -x -k "guid or c5" 2>&1 | tail -8
cd /tmp
RAW=/tmp/ddnm_prof; rm -rf $RAW; mkdir -p $RAW
B=32 timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $RAW/prof_cls -o cls -- python /root/repo/tools/cls_step.py 3 > /root/repo/gpurun_out/prof_cls32.log 2>&1
cd /root/repo
python tools/prof_summary.py $(find $RAW/prof_cls -name "*.db" | head -1) gpurun_out/r05_cls_b32_kernel_stats.md --after-marker finalize_psnr --forwards 3 > /dev/null; head -32 gpurun_out/r05_cls_b32_kernel_stats.md
