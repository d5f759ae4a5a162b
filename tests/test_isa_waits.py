"""The counted `s_waitcnt vmcnt(N)` waits of the LDS-DMA convolution kernels, checked against the ISSUE ORDER in the gfx950
ISA that hipcc actually emits (tools/isa_waits.py replays the request stream of every kernel instance's main loop).

The headline kernel (conv_igemm_f16.hip, split form) and the gather form keep two weight tiles in flight by LDS-DMA and wait
for the older one with an immediate that equals the number of requests issued behind it; a request the compiler (or an
edit) moves across a DMA request makes the wait return while the tile is still landing -- silently wrong numbers.  The
loops are written with a branch-free request stream so that the replay is exact; the last test moves one request in a
copy of the source and demands that the check fails.  CPU only (hipcc cross-compiles)."""
import os
import re
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import isa_waits  # noqa: E402

CSRC = isa_waits.CSRC


@pytest.fixture(scope="module")
def halo_asm(tmp_path_factory):
    return isa_waits.compile_isa(os.path.join(CSRC, "conv_igemm_f16.hip"), str(tmp_path_factory.mktemp("isa") / "f16.s"))


def test_halo_kernel_waits_match_the_issue_order(halo_asm):
    # the 8-wave instances (the 4-wave form of round 5, <2,2,4,2,..,W4>, waits with vmcnt(0) everywhere: nothing counted)
    res = isa_waits.analyse(halo_asm, r"conv3x3_halo_f16_kernelILi4ELi2ELi2ELi2E", min_barriers=9, depth=2)
    # <4,2,2,2, SRC16, SPLIT, ASCALE>: fp16 operands (fp32 source), fp16 source, split, split + operand scale
    assert len(res) == 4, list(res)
    for name, r in res.items():
        assert r["barriers"] == 9, (name, r)                         # one barrier per tap, taps unrolled
        assert r["groups"] == [2] * 9, (name, r)                     # BR = 2 LDS-DMA requests per wave and weight tile
        assert len(r["waits"]) == 9, (name, r)
        for tap, (n, behind) in enumerate(r["waits"]):
            assert n == behind, f"{name}: wait #{tap} has vmcnt({n}) but {behind} requests were issued behind the tile"
    # the split instances: HR = 6 halo slots in two halves of 3 (+ 2 GroupNorm vectors with the first half)
    split = [r for n, r in res.items() if "Lb0ELb1E" in n]
    assert len(split) == 2
    for r in split:
        assert sorted(n for n, _ in r["waits"]) == [2, 2, 2, 2, 2, 5, 5, 7, 7]


def test_gather_kernel_waits_match_the_issue_order(tmp_path):
    asm = isa_waits.compile_isa(os.path.join(CSRC, "conv_gather_s16.hip"), str(tmp_path / "gs.s"))
    res = isa_waits.analyse(asm, r"conv_gather_s16_kernel", min_barriers=4, depth=2)
    assert len(res) == 2
    for name, r in res.items():
        ar_br = {"Li2ELi2ELi2ELi2E": (4, 4), "Li2ELi2ELi1ELi1E": (2, 2)}[re.search(r"kernelI(\w+?)Ev", name).group(1)]
        for n, behind in r["waits"]:
            assert n == behind == ar_br[0] + 2 + ar_br[1], (name, r)   # AR gathers + 2 GroupNorm vectors + BR DMA requests
        assert set(r["groups"]) == {ar_br[1]}


def test_the_check_fails_when_a_request_is_moved(tmp_path):
    """Mutation: request the next chunk's first halo half BEFORE tap 0's weight-tile DMA instead of behind it.  The wait of
    tap 2 (vmcnt(7)) then leaves tap 0's tile among the 7 youngest requests: the replay must flag it."""
    src = open(os.path.join(CSRC, "conv_igemm_f16.hip")).read()
    issue = "                issue_step(step + 2, cur >= 1 ? cur - 1 : NWB - 1);"
    moved = "                if (tap == 0) prefetch_halo_part(nchunk, 0, HSPLIT, more);\n"
    assert src.count(issue) == 1 and src.count(moved) == 1
    mut = src.replace(moved, "")
    mut = mut.replace(issue, "                if (tap == 0) prefetch_halo_part(more ? chunk + 1 : chunk, 0, HSPLIT, more);\n"
                             "                asm volatile(\"\" ::: \"memory\");\n                __builtin_amdgcn_sched_barrier(0);\n" + issue)
    p = tmp_path / "conv_igemm_f16_mut.hip"
    p.write_text(mut)
    asm = isa_waits.compile_isa(str(p), str(tmp_path / "mut.s"), extra=["-I", CSRC])
    res = isa_waits.analyse(asm, r"conv3x3_halo_f16_kernelILi4ELi2ELi2ELi2ELb0ELb1ELb0E", min_barriers=9, depth=2)
    (r,) = res.values()
    bad = [(n, behind) for n, behind in r["waits"] if n != behind]
    assert bad and any(n > behind for n, behind in bad), r


def test_conv16_k_loops_issue_only_lds_dma(tmp_path):
    """conv16's request stream is data dependent (the halo pieces a wave fetches, the tail of a split-K slice), so its
    counted waits follow run-time bookkeeping (`pend` / `ahead`) instead of compile-time constants.  What that bookkeeping
    needs from the compiler: every request inside a K loop is an LDS-DMA (side-effecting: LLVM keeps them in program
    order, weight tile last) and none is a register load, which hipcc is free to sink across a DMA request -- and the
    immediates are 0, one or two request groups (GRP = halo groups per wave + 4 weight requests)."""
    asm = isa_waits.compile_isa(os.path.join(CSRC, "conv16.hip"), str(tmp_path / "c16.s"))
    res = isa_waits.dma_only_loops(asm, r"conv16_kernel")
    assert len(res) == 6, list(res)
    grp = {"Li9ELi1ELi1E": None, "Li9ELi4ELi4E": None, "Li9ELi2ELi4E": 4, "Li1ELi4ELi4E": None, "Li1ELi2ELi4E": 6, "Li1ELi1ELi4E": 5}
    for name, (n_loops, n_dma, n_reg, imm) in res.items():
        key = re.search(r"kernelI(\w+?)Ev", name).group(1)
        assert n_loops >= 1 and n_dma > 0, (name, n_loops, n_dma)
        assert n_reg == 0, f"{name}: {n_reg} register load(s) inside a K loop"
        allowed = {0} if grp[key] is None else {0, grp[key], 2 * grp[key]}
        assert set(imm) <= allowed, (name, imm, allowed)


# ---- persistent split-fp16 kernel (round 6): all-paths replay over the control-flow graph -------------------------------
def _p_wait(kind, tap, has_res, ascale):
    """Python restatement of conv_s16_persist.hip::p_wait (kinds: 0 MID, 1 LAST, 2 FIRST)."""
    BR, HR, HSPLIT = 2, 6, 3
    spread = lambda t: 10 if t == 0 else (9 if t <= 6 else 0)        # noqa: E731

    def extra(t):
        if kind == 2:
            return spread(t) + (1 if t == 0 else 0)
        if kind == 1:
            return ((4 + (1 if ascale else 0)) if t == 0 else 0) + (spread(t) if has_res else 0)
        return 0
    base = BR + HSPLIT + 2 if tap in (1, 2) else (BR + HR - HSPLIT if tap in (4, 5) else BR)
    return base + (extra(tap - 1) if tap >= 1 else 0) + (extra(tap - 2) if tap >= 2 else 0)


def test_persistent_kernel_waits_match_the_issue_order_on_every_path(tmp_path):
    """conv3x3_s16_persist_kernel<ASCALE, HAS_RES>: tile loop, three chunk kinds, an inner loop and exec-mask branches --
    every counted wait in front of a barrier is checked by walking back from it along EVERY path of the kernel's CFG: the
    requests met up to the youngest request of the awaited weight tile must equal the immediate on all of them (deferred
    stores, residual / bias / bound loads and the statistics store included), and the immediates are the source's table."""
    asm = isa_waits.compile_isa(os.path.join(CSRC, "conv_s16_persist.hip"), str(tmp_path / "p.s"))
    res = isa_waits.analyse_cfg(asm, r"conv3x3_s16_persist_kernel", depth=2, group_size=2)
    assert len(res) == 5, list(res)          # <ASCALE, HAS_RES> x 4 + the fused-shortcut instance <true, false, true>
    for name, waits in res.items():
        m = re.search(r"kernelILb(\d)ELb(\d)ELb(\d)E", name)
        ascale, has_res, has_skip = (g == "1" for g in m.groups())
        tap_waits = [w for w in waits if w[0] > 0]
        if has_skip:
            # the shortcut phase (plain loads + its own full waits) sits between a tile's LAST chunk and the next FIRST chunk,
            # and this instance keeps two spilled LDS addresses in scratch: paths through them see MORE requests behind the
            # awaited tile than the immediate allows in flight -- an over-wait (VMEM retires in order), never a stale tile.
            # Demanded: no path with fewer, and the tight path exists.
            for n, found, line in tap_waits:
                assert min(found) == n, f"{name}: wait vmcnt({n}) at line {line}: paths = {found}"
            continue
        assert len(waits) == 27, (name, len(waits))                 # 3 chunk kinds x 9 taps, each exactly once in the code
        for n, found, line in waits:
            assert found == [n], f"{name}: wait vmcnt({n}) at line {line}: requests behind the awaited tile on the paths = {found}"
        want = sorted(_p_wait(kind, tap, has_res, ascale) for kind in (0, 1, 2) for tap in range(9))
        assert sorted(n for n, _, _ in waits) == want, (name, sorted(n for n, _, _ in waits), want)


def test_the_all_paths_check_fails_when_the_table_is_wrong(tmp_path):
    """Mutation: one deferred store fewer at tap 0 than the table says (the FIRST chunk's waits of taps 1 / 2 then allow one
    request too many in flight: the awaited weight tile may still be landing).  The replay must flag exactly those waits."""
    src = open(os.path.join(CSRC, "conv_s16_persist.hip")).read()
    old = "            for (int k = p_first_of(tap); k < p_first_of(tap + 1); ++k)\n                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(outv[k]), r_out,"
    assert src.count(old) == 1
    mut = src.replace(old, "            for (int k = p_first_of(tap) + (tap == 0 ? 1 : 0); k < p_first_of(tap + 1); ++k)\n"
                           "                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(outv[k]), r_out,")
    path = os.path.join(CSRC, "_mut_persist_test.hip")           # next to the headers it includes
    try:
        open(path, "w").write(mut)
        asm = isa_waits.compile_isa(path, str(tmp_path / "m.s"))
    finally:
        os.remove(path)
    res = isa_waits.analyse_cfg(asm, r"conv3x3_s16_persist_kernelILb0ELb0E", depth=2, group_size=2)
    (waits,) = res.values()
    bad = [(n, found) for n, found, _ in waits if found != [n]]
    assert len(bad) == 2 and all(found == [n - 1] for n, found in bad), bad
