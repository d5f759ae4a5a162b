"""Replay the vector-memory request stream of the LDS-DMA kernels from their gfx950 ISA and check the counted waits.

The split-fp16 / fp16 convolution kernels (ddnm_amd/csrc/conv_igemm_f16.hip, conv_gather_s16.hip, conv16.hip) keep
several weight tiles in flight by LDS-DMA (`buffer_load_dwordx4 ... lds`) and wait for the OLDEST one with
`s_waitcnt vmcnt(N)`, N = the number of requests issued behind it.  VMEM requests retire in order, so the immediate is
correct exactly when N does not exceed the number of requests that were issued after the last request of the awaited tile --
a property of the ISSUE ORDER hipcc chose, not of the source.  This module compiles a source with `hipcc -S`, finds the
main loop of each kernel instance and replays its request stream:

  * every request (`buffer_load* / global_load*`), every `s_waitcnt vmcnt` and every `s_barrier` of the loop must be
    executed unconditionally (no forward branch may jump over one): the stream is then the same on every iteration and
    can be replayed from the text;
  * at every counted wait in front of a barrier, the tile read behind that barrier is the one requested `depth` DMA groups
    ago (a group = the consecutive `... lds` requests of one step); the replay counts the requests issued behind the last
    request of that group and demands  N == that count  (`<` would be a wasted wait, `>` a stale LDS tile).

Used by tests/test_isa_waits.py (which also checks that moving a request in the source makes the check FAIL).
"""
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "ddnm_amd", "csrc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "--offload-device-only", "-S"]


def hipcc():
    import shutil
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found")


def compile_isa(src_path, out_path, extra=()):
    r = subprocess.run([hipcc()] + FLAGS + list(extra) + ["-o", out_path, src_path], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(r.stderr[-2000:])
    return open(out_path).read()


def kernels(asm):
    """mangled name -> list of instruction lines of every kernel in the assembly text."""
    out, name, body = {}, None, []
    for line in asm.splitlines():
        m = re.match(r"^(_Z\w+):", line)
        if m:
            name, body = m.group(1), []
            out[name] = body
            continue
        if name is None:
            continue
        if line.startswith("\t.end_amdhsa_kernel") or line.startswith(".Lfunc_end"):
            name = None
            continue
        body.append(line)
    return out


_LABEL = re.compile(r"^(\.LBB\d+_\d+):")
_BRANCH = re.compile(r"^\ts_c?branch\w*\s+(\.LBB\d+_\d+)")
_REQ = re.compile(r"^\t(buffer_load|global_load|buffer_store|global_store|buffer_atomic|global_atomic|scratch_)")
_WAIT = re.compile(r"^\ts_waitcnt\b(.*)")
_VM = re.compile(r"vmcnt\((\d+)\)")


def main_loop(lines, min_barriers):
    """(start, end) line indices of the innermost backward-branch region with the most MFMAs and >= min_barriers barriers."""
    labels = {m.group(1): i for i, l in enumerate(lines) for m in [_LABEL.match(l)] if m}
    best = None
    for i, l in enumerate(lines):
        m = _BRANCH.match(l)
        if not m or m.group(1) not in labels or labels[m.group(1)] >= i:
            continue
        a, b = labels[m.group(1)], i
        region = lines[a:b + 1]
        nb = sum(1 for x in region if x.startswith("\ts_barrier"))
        nm = sum(1 for x in region if "v_mfma" in x)
        if nb >= min_barriers and nm > 0:
            # prefer the smallest region that still holds the barriers (innermost loop)
            key = (b - a)
            if best is None or key < best[0]:
                best = (key, a, b)
    if best is None:
        raise AssertionError("no loop with counted waits found")
    return best[1], best[2]


def events(lines, a, b):
    """Linear event list of the loop body + the structural check that no request / wait / barrier sits in the shadow of a
    forward branch (= is executed conditionally)."""
    labels = {m.group(1): i for i, l in enumerate(lines) for m in [_LABEL.match(l)] if m}
    shadows = []
    for i in range(a, b):
        m = _BRANCH.match(lines[i])
        if m and m.group(1) in labels and i < labels[m.group(1)] <= b:
            shadows.append((i, labels[m.group(1)]))
        # (a branch to a label outside [a, b] is a loop exit: it ends the replayed stream, it does not reorder it)
    ev = []
    for i in range(a, b + 1):
        l = lines[i]
        kind = None
        if _REQ.match(l):
            kind = ("dma" if re.search(r"\blds\b", l) else "req", l.strip())
        elif l.startswith("\ts_barrier"):
            kind = ("barrier", None)
        else:
            w = _WAIT.match(l)
            if w and _VM.search(w.group(1)):
                # the source's counted waits are `vmcnt(N) lgkmcnt(0)`; waits the compiler adds for registers are plain
                kind = ("wait", (int(_VM.search(w.group(1)).group(1)), "lgkmcnt(0)" in w.group(1)))
        if kind is None:
            continue
        if kind[0] == "wait" and not kind[1][1]:
            continue        # a register wait the compiler placed (it can only make the queue shorter): not part of the check
        for (s, t) in shadows:
            if s < i < t:
                raise AssertionError(f"{kind[0]} at line {i} is conditionally executed (branch at {s} jumps to {t}): "
                                     f"the request stream is not uniform: {l.strip()}")
        ev.append(kind)
    return ev


def check_loop(ev, depth):
    """Replay the cyclic event list.  For every counted wait of the source (`vmcnt(N) lgkmcnt(0)` directly in front of a
    barrier) returns (N, requests issued behind the last request of the DMA group `depth` groups back)."""
    n = len(ev)
    seq = ev + ev + ev                      # cyclic: analyse the middle copy with a full history behind it
    results = []
    for idx in range(n, 2 * n):
        kind, val = seq[idx]
        if kind != "wait" or not val[1]:
            continue
        if idx + 1 >= len(seq) or seq[idx + 1][0] != "barrier":
            continue
        behind, found, in_group = 0, [], False
        j = idx - 1
        while j >= 0 and len(found) < depth:
            k = seq[j][0]
            if k == "dma":
                if not in_group:            # youngest request of a group: everything counted so far was issued behind it
                    found.append(behind)
                    in_group = True
                behind += 1
            elif k == "req":                # (a register load between the DMA requests of one step is older than the
                behind += 1                 #  group's youngest request: it only counts for the groups before it)
            elif k == "barrier":
                in_group = False
            j -= 1
        if len(found) < depth:
            raise AssertionError("could not find the awaited DMA group")
        results.append((val[0], found[depth - 1]))
    if not results:
        raise AssertionError("no counted wait in front of a barrier found")
    return results


def group_sizes(ev):
    sizes, cur = [], 0
    for k, _ in ev:
        if k == "dma":
            cur += 1
        elif k == "barrier" and cur:
            sizes.append(cur)
            cur = 0
    if cur:
        sizes.append(cur)
    return sizes


def analyse(asm, name_re, min_barriers, depth):
    """All kernel instances whose mangled name matches: {name: [(N, behind), ...]}."""
    out = {}
    for name, lines in kernels(asm).items():
        if not re.search(name_re, name):
            continue
        a, b = main_loop(lines, min_barriers)
        ev = events(lines, a, b)
        out[name] = dict(waits=check_loop(ev, depth), groups=group_sizes(ev),
                         barriers=sum(1 for k, _ in ev if k == "barrier"))
    if not out:
        raise AssertionError(f"no kernel matches {name_re}")
    return out


if __name__ == "__main__":
    import sys
    import tempfile
    src, name_re, nb, depth = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
    with tempfile.TemporaryDirectory() as td:
        asm = compile_isa(os.path.join(CSRC, src), os.path.join(td, "k.s"), sys.argv[5:])
    for k, v in analyse(asm, name_re, nb, depth).items():
        print(k, v)


def k_loops(lines):
    """Innermost backward-branch regions that hold MFMAs, a barrier, LDS-DMA requests and no store: the K loops."""
    labels = {m.group(1): i for i, l in enumerate(lines) for m in [_LABEL.match(l)] if m}
    loops = []
    for i, l in enumerate(lines):
        m = _BRANCH.match(l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            a, b = labels[m.group(1)], i
            reg = lines[a:b + 1]
            if any("v_mfma" in x for x in reg) and any(x.startswith("\ts_barrier") for x in reg) \
                    and any(_REQ.match(x) and re.search(r"\blds\b", x) for x in reg) \
                    and not any(re.match(r"\t(global|buffer)_store", x) for x in reg):
                loops.append((a, b))
    return [(a, b) for (a, b) in loops if not any((c, d) != (a, b) and a <= c and d <= b for (c, d) in loops)]


def dma_only_loops(asm, name_re):
    """For kernels whose request stream is data dependent (conv16: the halo pieces a wave fetches, the tail of a slice) the
    counted waits are kept correct by run-time bookkeeping (`pend` / `ahead`) that assumes ONE thing about the ISA: every
    request inside the K loops is an LDS-DMA -- a side-effecting instruction LLVM keeps in program order -- and never a
    register load, which the compiler is free to move across a DMA request (the hazard the halo kernel's replay guards
    against).  Returns {kernel: (number of K loops, DMA requests, register loads, sorted vmcnt immediates)}."""
    out = {}
    for name, lines in kernels(asm).items():
        if not re.search(name_re, name):
            continue
        loops = k_loops(lines)
        n_dma = n_reg = 0
        imm = set()
        for a, b in loops:
            for l in lines[a:b + 1]:
                if _REQ.match(l):
                    if re.search(r"\blds\b", l):
                        n_dma += 1
                    else:
                        n_reg += 1
                w = _WAIT.match(l)
                if w and _VM.search(w.group(1)):
                    imm.add(int(_VM.search(w.group(1)).group(1)))
        out[name] = (len(loops), n_dma, n_reg, sorted(imm))
    return out


# ---------------------------------------------------------------------------------------------------------------------------
# All-paths replay over the control-flow graph (round 6: conv_s16_persist.hip).  The persistent kernel's request stream runs
# through a tile loop with three kinds of chunks (FIRST / MID / LAST), an inner loop (MID) and small exec-mask branches
# (ds_write guards), so there is no single loop body to replay cyclically.  Here every counted wait in front of a barrier
# is checked by walking BACKWARDS from it along EVERY path of the kernel's CFG until `depth` DMA groups (= `group_size`
# LDS-DMA requests each) have been passed: the number of requests met on the way up to the youngest request of the awaited
# group must equal the wait's immediate on all of them.
def _blocks(lines):
    """Basic blocks: (start, end_exclusive) index ranges, label -> block id, successors per block."""
    starts = {0}
    for i, l in enumerate(lines):
        if _LABEL.match(l):
            starts.add(i)
        if _BRANCH.match(l) or l.startswith("\ts_endpgm"):
            starts.add(i + 1)
    starts = sorted(s for s in starts if s < len(lines))
    blocks = [(s, (starts[k + 1] if k + 1 < len(starts) else len(lines))) for k, s in enumerate(starts)]
    label_of = {}
    for b, (s, e) in enumerate(blocks):
        m = _LABEL.match(lines[s])
        if m:
            label_of[m.group(1)] = b
    succ = [[] for _ in blocks]
    for b, (s, e) in enumerate(blocks):
        last = lines[e - 1] if e > s else ""
        m = _BRANCH.match(last)
        if last.startswith("\ts_endpgm"):
            continue
        if m and m.group(1) in label_of:
            succ[b].append(label_of[m.group(1)])
            if not last.startswith("\ts_branch"):         # conditional: falls through as well
                if b + 1 < len(blocks):
                    succ[b].append(b + 1)
        elif b + 1 < len(blocks):
            succ[b].append(b + 1)
    return blocks, succ


def _block_events(lines, s, e):
    ev = []
    for i in range(s, e):
        l = lines[i]
        if _REQ.match(l):
            ev.append(("dma" if re.search(r"\blds\b", l) else "req", i))
        elif l.startswith("\ts_barrier"):
            ev.append(("barrier", i))
        else:
            w = _WAIT.match(l)
            if w and _VM.search(w.group(1)):
                ev.append(("wait", (int(_VM.search(w.group(1)).group(1)), "lgkmcnt(0)" in w.group(1), i)))
    return ev


def analyse_cfg(asm, name_re, depth, group_size):
    """{kernel: [(N, sorted set of request counts found behind the awaited DMA group over all paths, line), ...]} for every
    `s_waitcnt vmcnt(N) lgkmcnt(0)` (N > 0) that directly precedes an s_barrier.  A count ABOVE N on some path is an
    over-wait there (safe: VMEM retires in order), a count BELOW N a stale tile; 64 stands for "more than vmcnt counts"."""
    out = {}
    for name, lines in kernels(asm).items():
        if not re.search(name_re, name):
            continue
        blocks, succ = _blocks(lines)
        pred = [[] for _ in blocks]
        for b, ss in enumerate(succ):
            for t in ss:
                pred[t].append(b)
        evs = [_block_events(lines, s, e) for (s, e) in blocks]
        results = []
        for b, ev in enumerate(evs):
            for k, (kind, val) in enumerate(ev):
                if kind != "wait" or not val[1] or val[0] == 0:
                    continue
                # directly in front of a barrier (same block, next event)
                if k + 1 >= len(ev) or ev[k + 1][0] != "barrier":
                    continue
                found, seen = set(), set()
                stack = [(b, k - 1, 0, 0, 0)]          # (block, event index, groups passed, dmas in the open group, behind)
                while stack:
                    st = stack.pop()
                    if st in seen:
                        continue
                    seen.add(st)
                    bb, idx, g, ing, behind = st
                    done = False
                    while idx >= 0:
                        kd = evs[bb][idx][0]
                        if kd == "dma":
                            if ing == 0:                 # youngest request of a group: everything counted so far is behind it
                                g += 1
                                if g == depth:
                                    found.add(behind)
                                    done = True
                                    break
                            ing = (ing + 1) % group_size
                            behind += 1
                        elif kd == "req":
                            behind += 1
                        idx -= 1
                    if done:
                        continue
                    if behind > 64:
                        # more requests than vmcnt can count (a loop of plain loads between the DMA groups, e.g. the fused
                        # shortcut phase of the persistent kernel): recorded as "> 63", i.e. an over-wait on that path
                        found.add(64)
                        continue
                    for pb in pred[bb]:
                        stack.append((pb, len(evs[pb]) - 1, g, ing, behind))
                    # (a path that reaches the kernel entry without `depth` groups does not exist for these kernels: the
                    # prologue requests two weight tiles before the first wait)
                results.append((val[0], sorted(found), val[2]))
        out[name] = results
    if not out:
        raise AssertionError(f"no kernel matches {name_re}")
    return out
