"""Build libddnm_hip.so (gfx950) in-tree with hipcc.  No GPU needed to compile."""
import hashlib
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libddnm_hip.so")
STAMP = os.path.join(HERE, ".libddnm_hip.stamp")
OBJ = os.path.join(HERE, "_obj")
SOURCES = ["conv_igemm_f32.hip", "conv_igemm_f16.hip", "conv_s16_persist.hip", "conv_gather_s16.hip", "conv_small_f32.hip", "conv1x1_f16.hip", "conv16.hip", "act16.hip", "attn16.hip", "attn_d512.hip", "attn16_bwd.hip", "gemm_f32.hip", "groupnorm.hip", "misc.hip", "ddnm_step.hip", "fwht.hip", "backward.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wno-unused-function"]


def _hipcc():
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (set HIPCC)")


def _digest():
    h = hashlib.sha256()
    for f in sorted(os.listdir(CSRC)) + ["../../include/ddnm_hip.h"]:
        with open(os.path.join(CSRC, f), "rb") as fh:
            h.update(f.encode() + fh.read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


LAST_BUILD = {"compiled": False}      # whether the last build() call ran hipcc (reported by __graft_entry__.build)


def build(force=False, verbose=False):
    """Compile libddnm_hip.so unless the stamp says it already matches the sources.  The digest is also compiled
    INTO the binary (ddnm_build_digest()), which is what the loader trusts."""
    dig = _digest()
    LAST_BUILD["compiled"] = False
    if not force and os.path.exists(LIB) and os.path.exists(STAMP) and open(STAMP).read() == dig:
        return LIB
    LAST_BUILD["compiled"] = True
    tmp = LIB + f".tmp{os.getpid()}"
    # one object per source, compiled in parallel and cached under ddnm_amd/_obj/ by the hash of (source, every header,
    # flags): editing one kernel recompiles one file.  Only misc.hip sees the digest (ddnm_build_digest()).
    os.makedirs(OBJ, exist_ok=True)
    hh = hashlib.sha256()
    for f in sorted(os.listdir(CSRC)) + ["../../include/ddnm_hip.h"]:
        if f.endswith(".h"):
            with open(os.path.join(CSRC, f), "rb") as fh:
                hh.update(f.encode() + fh.read())
    hh.update(" ".join(FLAGS).encode())
    cflags = [f for f in FLAGS if f != "-shared"]
    jobs, objs = [], []
    for s in SOURCES:
        h = hashlib.sha256(hh.digest())
        with open(os.path.join(CSRC, s), "rb") as fh:
            h.update(fh.read())
        extra = [f'-DDDNM_BUILD_DIGEST="{dig}"'] if s == "misc.hip" else []
        h.update(" ".join(extra).encode())
        obj = os.path.join(OBJ, f"{s}.{h.hexdigest()[:16]}.o")
        objs.append(obj)
        if force or not os.path.exists(obj):
            jobs.append((s, [_hipcc()] + cflags + extra + ["-c", os.path.join(CSRC, s), "-o", obj + f".tmp{os.getpid()}"], obj))
    if jobs:
        from concurrent.futures import ThreadPoolExecutor

        def run(job):
            s, cmd, obj = job
            if verbose:
                print(" ".join(cmd), flush=True)
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode == 0:
                os.replace(cmd[-1], obj)
            return s, r
        with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 4)) as ex:
            results = list(ex.map(run, jobs))
        bad = [(s, r) for s, r in results if r.returncode != 0]
        for s, r in results:
            if r.returncode != 0 or (verbose and r.stderr):
                sys.stderr.write(r.stdout + r.stderr)
        if bad:
            raise RuntimeError("hipcc failed: " + ", ".join(s for s, _ in bad))
    keep = set(objs)
    for f in os.listdir(OBJ):                       # stale objects of earlier source states
        if os.path.join(OBJ, f) not in keep and f.endswith(".o"):
            os.remove(os.path.join(OBJ, f))
    cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", tmp]
    if verbose:
        print(" ".join(cmd))
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stdout + r.stderr)
        raise RuntimeError("hipcc (link) failed")
    os.replace(tmp, LIB)               # atomic: a concurrent loader never maps a half-written file
    with open(STAMP, "w") as f:
        f.write(dig)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
