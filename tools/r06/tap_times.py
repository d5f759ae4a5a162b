#!/usr/bin/env python
"""Per-tap timeline of the persistent split-fp16 kernel (development tool, round 6): an instrumented COPY of
conv_s16_persist.hip (tools/_build/) stamps s_memrealtime (100 MHz) in front of every tap's counted wait and behind its
barrier, for waves 0 and 4 of every workgroup's SECOND tile.  Prints, per chunk kind and tap, the mean time a wave spends
waiting (wait + barrier) and working (barrier -> next wait).     python tools/r06/tap_times.py [--build-only]"""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
CSRC = os.path.join(ROOT, "ddnm_amd", "csrc")
OUT = os.path.join(ROOT, "tools", "_build")


def build():
    os.makedirs(OUT, exist_ok=True)
    src = open(os.path.join(CSRC, "conv_s16_persist.hip")).read()
    src = src.replace('typedef unsigned u32x4 __attribute__((ext_vector_type(4)));',
                      'typedef unsigned u32x4 __attribute__((ext_vector_type(4)));\n'
                      '__device__ unsigned long long* g_dbg;\n'
                      'extern "C" void ddnm_dbg_set(void* p) { (void)hipMemcpyToSymbol(HIP_SYMBOL(g_dbg), &p, sizeof(p)); }\n', 1)
    # stamps go to LDS (a global store per stamp would join the counted request stream and make the waits over-wait):
    # slot = ((chunk * 9 + tap) * 2 + {0: before the wait, 1: behind the barrier}) of waves 0 / 4, dumped after the tile
    src = src.replace('__shared__ __attribute__((aligned(1024))) char lds_all[NWB * WTILE + P_HBYTES + WM * BN * 2 * 4];',
                      '__shared__ __attribute__((aligned(1024))) char lds_all[NWB * WTILE + P_HBYTES + WM * BN * 2 * 4 + 2 * 16 * 9 * 2 * 8];\n'
                      '    unsigned long long* const dbg_lds = reinterpret_cast<unsigned long long*>(lds_all + NWB * WTILE + P_HBYTES + WM * BN * 2 * 4);', 1)
    src = src.replace('        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(p_wait(kind, tap, HAS_RES, ASCALE)) : "memory");\n        __builtin_amdgcn_s_barrier();\n',
                      '        const bool dbg_on = g_dbg && tile_no == 1 && (tid == 0 || tid == 256);\n'
                      '        unsigned long long* dbg_p = dbg_lds + ((tid >> 8) * 16 * 9 + (chunk * 9 + tap)) * 2;\n'
                      '        if (dbg_on) dbg_p[0] = wall_clock64();\n'
                      '        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(p_wait(kind, tap, HAS_RES, ASCALE)) : "memory");\n        __builtin_amdgcn_s_barrier();\n'
                      '        if (dbg_on) dbg_p[1] = wall_clock64();\n', 1)
    src = src.replace('    bool pending = false;\n    for (;;) {', '    bool pending = false;\n    for (;; ++tile_no) {', 1)
    src = src.replace('    int hb = 0, n_img_next = 0;', '    int hb = 0, n_img_next = 0, tile_no = 0;', 1)
    src = src.replace('    // ---- the last tile\'s values leave directly\n    __syncthreads();',
                      '    __syncthreads();\n    if (g_dbg) for (int i = tid; i < 2 * 16 * 9 * 2; i += NTHREADS) g_dbg[(size_t)blockIdx.x * 2 * 16 * 9 * 2 + i] = dbg_lds[i];\n'
                      '    // ---- the last tile\'s values leave directly\n    __syncthreads();', 1)
    assert "g_dbg[(size_t)blockIdx.x" in src
    assert src.count("dbg_p[") == 2 and "++tile_no" in src
    tmp = os.path.join(OUT, "taps_src")
    os.makedirs(tmp, exist_ok=True)
    for f in os.listdir(CSRC):
        if f.endswith(".h"):
            open(os.path.join(tmp, f), "w").write(open(os.path.join(CSRC, f)).read().replace('"../../include/ddnm_hip.h"', f'"{ROOT}/include/ddnm_hip.h"'))
    f16 = open(os.path.join(CSRC, "conv_igemm_f16.hip")).read()
    open(os.path.join(tmp, "conv_igemm_f16.hip"), "w").write(f16)
    open(os.path.join(tmp, "conv_s16_persist.hip"), "w").write(src)
    so = os.path.join(OUT, "libs16_taps.so")
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-value",
                    os.path.join(tmp, "conv_igemm_f16.hip"), os.path.join(tmp, "conv_s16_persist.hip"), "-o", so], check=True)
    return so


def main():
    so = build()
    if "--build-only" in sys.argv:
        return
    import numpy as np
    import torch
    import s16_probe as sp
    from ddnm_amd import ops
    from ddnm_amd._lib import ConvDesc
    lib = ctypes.CDLL(so)
    lib.ddnm_conv3x3_s16_f32.restype = ctypes.c_int32
    lib.ddnm_conv3x3_s16_f32.argtypes = [ctypes.POINTER(ConvDesc), ctypes.c_void_p]
    lib.ddnm_dbg_set.argtypes = [ctypes.c_void_p]
    dev = "cuda"
    stream = torch.cuda.current_stream().cuda_stream
    names = os.environ.get("SHAPES", "c128_128_256_gn_res,c256cat_128_256_gn").split(",")
    for s in sp.SHAPES:
        if s[0] not in names:
            continue
        name, B, C0, C1, Cout, H, ups, gn, res, skip = s
        t = sp.make(*s)
        Ho = t["Ho"]
        scale = ops.s16_weight_scale(t["w"])
        wp = ops.pack_conv_weight_s16(t["w"], scale)
        out = torch.empty(B, Ho, Ho, Cout, device=dev)
        stats = torch.empty(B * 1024 * Cout * 2, device=dev)
        d = ConvDesc()
        d.src0, d.src1, d.weight, d.bias = t["a"].data_ptr(), (t["b"].data_ptr() if C1 else None), wp.data_ptr(), t["bias"].data_ptr()
        d.res = t["r"].data_ptr() if res else None
        d.gn_scale, d.gn_shift = (t["sc"].data_ptr(), t["sh"].data_ptr()) if gn else (None, None)
        d.out, d.stats_out = out.data_ptr(), stats.data_ptr()
        d.B, d.Hin, d.Win, d.C0, d.C1, d.Cout = B, Ho, Ho, C0, C1, Cout
        d.ksize, d.stride, d.pad, d.Ho, d.Wo = 3, 1, 1, Ho, Ho
        d.ups, d.gn_silu, d.acc_scale = ups, 1, 1.0 / scale
        nch = (C0 + C1) // 32
        dbg = torch.zeros(256 * 2 * 16 * 9 * 2, dtype=torch.int64, device=dev)
        for _ in range(5):
            assert lib.ddnm_conv3x3_s16_f32(ctypes.byref(d), stream) == 0
        torch.cuda.synchronize()
        lib.ddnm_dbg_set(dbg.data_ptr())
        lib.ddnm_conv3x3_s16_f32(ctypes.byref(d), stream)
        torch.cuda.synchronize()
        lib.ddnm_dbg_set(None)
        a = dbg.cpu().numpy().reshape(256, 2, 16, 9, 2).astype(np.float64) * 0.01      # us
        a = a[:, :, :nch]
        flat = a.reshape(256, 2, nch * 9, 2)
        wait = flat[..., 1] - flat[..., 0]                          # wait + barrier of each tap
        work = flat[:, :, 1:, 0] - flat[:, :, :-1, 1]               # barrier -> the next tap's wait
        print(f"{name}: second tile of every workgroup, {nch} chunks; mean over 256 workgroups, us  [tap: wait | work]   (wave 0 / wave 4)")
        for c in range(nch):
            row = []
            for tp in range(9):
                k = c * 9 + tp
                w0, w4 = wait[:, 0, k].mean(), wait[:, 1, k].mean()
                if k < nch * 9 - 1:
                    x0, x4 = work[:, 0, k].mean(), work[:, 1, k].mean()
                else:
                    x0 = x4 = float("nan")
                row.append(f"{tp}: {w0:.2f}/{w4:.2f} | {x0:.2f}/{x4:.2f}")
            print(f"  chunk {c} ({'FIRST' if c == 0 else ('LAST' if c == nch - 1 else 'MID')}): " + "   ".join(row))
        tot = flat[:, 0, -1, 1] - flat[:, 0, 0, 0]
        print(f"  taps 0..{nch * 9 - 1}: {tot.mean():.2f} us; sum wait {wait[:, 0].sum(1).mean():.2f} (wave 0) {wait[:, 1].sum(1).mean():.2f} (wave 4); ideal MFMA time "
              f"{(nch * 9 - 1) * 1536 / 1.92e3:.2f} us at 1.92 GHz", flush=True)
        del t
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
