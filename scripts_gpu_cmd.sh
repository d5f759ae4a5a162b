cd /root/repo/ddnm_amd/csrc
S="conv_igemm_f32.hip conv_igemm_f16.hip gemm_f32.hip groupnorm.hip misc.hip ddnm_step.hip fwht.hip backward.hip"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-function -DDDNM_PROBE_NO_RES_FOLD $S -o /tmp/lib_nofold.so 2>/dev/null
cp ../libddnm_hip.so /tmp/lib_fold.so
cd /root/repo
for v in fold nofold fold nofold fold nofold; do
  cp /tmp/lib_$v.so ddnm_amd/libddnm_hip.so
  echo -n "$v "; python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'])"
done
