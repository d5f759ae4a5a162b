#!/usr/bin/env python
"""Where does a launch of the split-fp16 3x3 kernel spend its time OUTSIDE the tap loop?  (development tool, round 6)

Builds an instrumented copy of ddnm_amd/csrc/conv_igemm_f16.hip under tools/_build/ (the product sources are not touched):
wave 0 of every workgroup stamps s_memrealtime (100 MHz) at kernel entry, in front of the chunk loop (first halo staged),
behind it, behind the epilogue's last store instruction and behind the final drain, plus HW_ID / XCC_ID.  The script
prints per layer shape the mean phase lengths and, per CU, the gap between one workgroup's end and the next one's entry.

    python tools/r06/tile_phases.py [--build-only]
"""
import ctypes
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
CSRC = os.path.join(ROOT, "ddnm_amd", "csrc")
OUT = os.path.join(ROOT, "tools", "_build")


def build():
    os.makedirs(OUT, exist_ok=True)
    src = open(os.path.join(CSRC, "conv_igemm_f16.hip")).read()
    stamp = 'DBG_STAMP(%d);'
    src = src.replace('typedef unsigned u32x4 __attribute__((ext_vector_type(4)));',
                      'typedef unsigned u32x4 __attribute__((ext_vector_type(4)));\n'
                      '__device__ unsigned long long* g_dbg;\n'
                      'extern "C" void ddnm_dbg_set(void* p) { (void)hipMemcpyToSymbol(HIP_SYMBOL(g_dbg), &p, sizeof(p)); }\n'
                      '#define DBG_STAMP(i) do { if (threadIdx.x == 0 && g_dbg) g_dbg[(blockIdx.x + gridDim.x * blockIdx.y) * 8 + (i)] = wall_clock64(); } while (0)\n', 1)
    # entry
    src = src.replace('    const ddnm_conv_desc& d = p.d;\n    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;\n    const int wm = wave / WN, wn = wave % WN;\n    const int tile_id = xcd_swizzle',
                      '    DBG_STAMP(0);\n    if (threadIdx.x == 0 && g_dbg) { g_dbg[(blockIdx.x + gridDim.x * blockIdx.y) * 8 + 6] = __builtin_amdgcn_s_getreg(63492); g_dbg[(blockIdx.x + gridDim.x * blockIdx.y) * 8 + 7] = __builtin_amdgcn_s_getreg(63508); }\n'
                      '    const ddnm_conv_desc& d = p.d;\n    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;\n    const int wm = wave / WN, wn = wave % WN;\n    const int tile_id = xcd_swizzle', 1)
    src = src.replace('        stage_halo_part(0, 0, HR);\n        int hb = 0, step0 = 0;', '        stage_halo_part(0, 0, HR);\n        DBG_STAMP(1);\n        int hb = 0, step0 = 0;', 1)
    src = src.replace('        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the tail requests\n        __syncthreads();',
                      '        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the tail requests\n        __syncthreads();\n        DBG_STAMP(2);', 1)
    src = src.replace('    conv_epilogue<WM, WN, MT, NT, (MT * NT <= 4)>(p, tm, n_tile, m_tile, slice, acc, stat_lds, epi_scale);\n}',
                      '    DBG_STAMP(3);\n    conv_epilogue<WM, WN, MT, NT, (MT * NT <= 4)>(p, tm, n_tile, m_tile, slice, acc, stat_lds, epi_scale);\n    DBG_STAMP(4);\n'
                      '    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");\n    DBG_STAMP(5);\n}', 1)
    assert src.count("DBG_STAMP(") == 7, src.count("DBG_STAMP(")
    tmp = os.path.join(OUT, "phases_src")
    os.makedirs(tmp, exist_ok=True)
    for f in os.listdir(CSRC):
        if f.endswith(".h"):
            open(os.path.join(tmp, f), "w").write(open(os.path.join(CSRC, f)).read().replace('"../../include/ddnm_hip.h"', f'"{ROOT}/include/ddnm_hip.h"'))
    open(os.path.join(tmp, "conv_igemm_f16.hip"), "w").write(src)
    so = os.path.join(OUT, "libs16_phases.so")
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-value",
                    os.path.join(tmp, "conv_igemm_f16.hip"), "-o", so], check=True)
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
    ws = torch.empty(64 << 20, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    names = os.environ.get("SHAPES", "c128_128_256_gn_res,c128_128_256_plain,c256cat_128_256_gn,c128_128_128_gn_res,c256_256_64_gn_res,c128_128_up256").split(",")
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
        amax = torch.full((B, 32), 8.0, device=dev)
        d = ConvDesc()
        d.src0, d.src1, d.weight, d.bias = t["a"].data_ptr(), (t["b"].data_ptr() if C1 else None), wp.data_ptr(), t["bias"].data_ptr()
        d.res = t["r"].data_ptr() if res else None
        d.gn_scale, d.gn_shift = (t["sc"].data_ptr(), t["sh"].data_ptr()) if gn else (None, None)
        d.out, d.stats_out = out.data_ptr(), stats.data_ptr()
        d.B, d.Hin, d.Win, d.C0, d.C1, d.Cout = B, Ho, Ho, C0, C1, Cout
        d.ksize, d.stride, d.pad, d.Ho, d.Wo = 3, 1, 1, Ho, Ho
        d.ups, d.gn_silu, d.acc_scale = ups, 1, 1.0 / scale
        d.amax_in = amax.data_ptr()
        d.workspace, d.workspace_floats = ws.data_ptr(), ws.numel()
        nwg = B * (Ho * Ho // 256) * (Cout // 128)
        dbg = torch.zeros(nwg * 16 * 8, dtype=torch.int64, device=dev)
        lib.ddnm_dbg_set(None)
        for _ in range(5):
            assert lib.ddnm_conv3x3_s16_f32(ctypes.byref(d), stream) == 0
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            lib.ddnm_conv3x3_s16_f32(ctypes.byref(d), stream)
        e1.record()
        torch.cuda.synchronize()
        us_plain = e0.elapsed_time(e1) * 100
        lib.ddnm_dbg_set(dbg.data_ptr())
        lib.ddnm_conv3x3_s16_f32(ctypes.byref(d), stream)
        torch.cuda.synchronize()
        lib.ddnm_dbg_set(None)
        a = dbg.cpu().numpy().reshape(-1, 8)
        a = a[a[:, 0] != 0]
        T = a[:, :6].astype(np.float64) * 0.01            # us
        t0 = T[:, 0].min()
        ph = {"entry->loop (prologue)": T[:, 1] - T[:, 0], "tap loop": T[:, 2] - T[:, 1], "shortcut phase": T[:, 3] - T[:, 2],
              "epilogue issue": T[:, 4] - T[:, 3], "store drain": T[:, 5] - T[:, 4], "workgroup total": T[:, 5] - T[:, 0]}
        cu = (a[:, 7] & 0xf) * 100000 + (a[:, 6] & ~0x3f & 0xffff)     # (XCC, SE / SH / CU bits of HW_ID)
        gaps, per_cu = [], []
        for c in np.unique(cu):
            rows = T[cu == c]
            rows = rows[np.argsort(rows[:, 0])]
            per_cu.append(len(rows))
            gaps += list(rows[1:, 0] - rows[:-1, 5])
        span = T[:, 5].max() - t0
        print(f"{name:24s} {us_plain:7.1f} us un-instrumented | instrumented span {span:7.1f} us, {len(a)} workgroups on {len(per_cu)} CU ids "
              f"({min(per_cu)}..{max(per_cu)} per CU)")
        for k, v in ph.items():
            print(f"    {k:26s} mean {v.mean():7.2f} us   p10 {np.percentile(v, 10):7.2f}   p90 {np.percentile(v, 90):7.2f}")
        g = np.array(gaps) if gaps else np.zeros(1)
        print(f"    {'end -> next entry (same CU)':26s} mean {g.mean():7.2f} us   p10 {np.percentile(g, 10):7.2f}   p90 {np.percentile(g, 90):7.2f}")
        first = T[:, 0] - t0
        print(f"    first-round entries: {np.sort(first)[:256].max():.2f} us after the first; last workgroup ends at {span:.1f} us", flush=True)
        del t
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
