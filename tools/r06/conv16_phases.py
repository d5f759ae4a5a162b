#!/usr/bin/env python
"""Where does a workgroup of conv16_kernel<9,4,4> (ADM fp16 torso) spend its time?  Instrumented COPY of conv16.hip
(tools/_build/): s_memrealtime stamps at entry, in front of the K loop, behind it, behind the epilogue.  Round 6 tool."""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
CSRC = os.path.join(ROOT, "ddnm_amd", "csrc"); OUT = os.path.join(ROOT, "tools", "_build")


def build():
    os.makedirs(OUT, exist_ok=True)
    s = open(os.path.join(CSRC, "conv16.hip")).read()
    s = s.replace('typedef __attribute__((address_space(3))) void lds_ptr_t;',
                  'typedef __attribute__((address_space(3))) void lds_ptr_t;\n__device__ unsigned long long* g_dbg;\n'
                  'extern "C" void ddnm_dbg_set(void* p) { (void)hipMemcpyToSymbol(HIP_SYMBOL(g_dbg), &p, sizeof(p)); }\n'
                  '#define DBG_STAMP(i) do { if (threadIdx.x == 0 && g_dbg) g_dbg[blockIdx.x * 8 + (i)] = wall_clock64(); } while (0)\n', 1)
    s = s.replace('    const ddnm_conv16_desc& d = p.d;\n    const int tid = threadIdx.x, lane = tid & 63;\n    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);\n    const int wm = WNW == 4',
                  '    DBG_STAMP(0);\n    if (threadIdx.x == 0 && g_dbg) { g_dbg[blockIdx.x * 8 + 6] = __builtin_amdgcn_s_getreg(63492); g_dbg[blockIdx.x * 8 + 7] = __builtin_amdgcn_s_getreg(63508); }\n'
                  '    const ddnm_conv16_desc& d = p.d;\n    const int tid = threadIdx.x, lane = tid & 63;\n    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);\n    const int wm = WNW == 4', 1)
    s = s.replace('#pragma unroll 1\n    for (int c = c_begin; c < c_end; ++c) {\n        const bool more = c + 1 < c_end;', 'DBG_STAMP(1);\n#pragma unroll 1\n    for (int c = c_begin; c < c_end; ++c) {\n        const bool more = c + 1 < c_end;', 1)
    s = s.replace('    if (TAPS == 9 && n_skip > 0) {\n        // fused 1x1 shortcut', '    DBG_STAMP(2);\n    if (TAPS == 9 && n_skip > 0) {\n        // fused 1x1 shortcut', 1)
    s = s.replace('    __syncthreads();                   // all fragment reads done: LDS becomes the epilogue\'s staging area', '    DBG_STAMP(3);\n    __syncthreads();                   // all fragment reads done: LDS becomes the epilogue\'s staging area', 1)
    # end of kernel: the statistics block closes the kernel; stamp in front of it and after a full drain
    s = s.replace('    if (d.stats_out && !partial16) {', '    DBG_STAMP(4);\n    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");\n    DBG_STAMP(5);\n    if (d.stats_out && !partial16) {', 1)
    assert s.count("DBG_STAMP(") == 7, s.count("DBG_STAMP(")
    tmp = os.path.join(OUT, "c16p_src"); os.makedirs(tmp, exist_ok=True)
    for f in os.listdir(CSRC):
        if f.endswith(".h"):
            open(os.path.join(tmp, f), "w").write(open(os.path.join(CSRC, f)).read().replace('"../../include/ddnm_hip.h"', f'"{ROOT}/include/ddnm_hip.h"'))
    open(os.path.join(tmp, "conv16.hip"), "w").write(s)
    so = os.path.join(OUT, "libc16_phases.so")
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-value", os.path.join(tmp, "conv16.hip"), "-o", so], check=True)
    return so


def main():
    so = build()
    if "--build-only" in sys.argv:
        return
    import numpy as np, torch
    from ddnm_amd import _lib, ops
    import conv16_bench as cb
    alt = ctypes.CDLL(so)
    for name in ("ddnm_conv16", "ddnm_conv16_supported", "ddnm_conv16_workspace_floats", "ddnm_conv16_stats_tiles", "ddnm_conv16_fuses_fin"):
        fn = getattr(alt, name); fn.restype, fn.argtypes = _lib.PROTOTYPES[name]
    alt.ddnm_dbg_set.argtypes = [ctypes.c_void_p]
    base = _lib.lib()

    class Mix:                                       # conv16 entry points from the instrumented copy, everything else from the product
        def __getattr__(self, n):
            return getattr(alt, n) if n.startswith("ddnm_conv16") else getattr(base, n)
    B = 4
    for name, cin, cout, H, k, res, ups, _ in cb.SHAPES:
        if k != 3 or H < 128:
            continue
        Hs = H // 2 if ups else H
        x16 = torch.randn(B, Hs, Hs, cin, device="cuda").half()
        w16 = ops.pack_conv_weight16(torch.randn(cout, cin, k, k, device="cuda") * (k * k * cin) ** -0.5)
        bias = torch.randn(cout, device="cuda")
        r16 = torch.randn(B, H, H, cout, device="cuda").half() if res else None
        sc, sh = torch.rand(B, cin, device="cuda") + 0.5, torch.randn(B, cin, device="cuda") * 0.2
        f = lambda: ops.conv16(x16, w16, cout, k, bias=bias, res=r16, ups=ups, gn=(sc, sh))      # noqa: E731
        _lib._lib = Mix()
        alt.ddnm_dbg_set(None)
        us = cb.timeit(f) * 1e3
        nwg = B * (H * H // 256) * ((cout + 255) // 256)
        dbg = torch.zeros(nwg * 8 + 64, dtype=torch.int64, device="cuda")
        alt.ddnm_dbg_set(dbg.data_ptr()); f(); torch.cuda.synchronize(); alt.ddnm_dbg_set(None)
        _lib._lib = base
        a = dbg.cpu().numpy()[:nwg * 8].reshape(-1, 8); a = a[a[:, 0] != 0]
        T = a[:, :6].astype(np.float64) * 0.01
        ph = {"entry -> K loop (first tiles + activation)": T[:, 1] - T[:, 0], "K loop": T[:, 2] - T[:, 1], "shortcut + residual request": T[:, 3] - T[:, 2],
              "epilogue issue": T[:, 4] - T[:, 3], "store drain": T[:, 5] - T[:, 4], "workgroup": T[:, 5] - T[:, 0]}
        print(f"{name:22s} {us:7.1f} us; {len(a)} workgroups; span {T[:, 5].max() - T[:, 0].min():.1f} us")
        for k2, v in ph.items():
            print(f"    {k2:44s} mean {v.mean():7.2f} us  p10 {np.percentile(v, 10):7.2f}  p90 {np.percentile(v, 90):7.2f}")


if __name__ == "__main__":
    main()
