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
SOURCES = ["conv_igemm_f32.hip", "conv_igemm_f16.hip", "conv_gather_s16.hip", "conv_small_f32.hip", "conv1x1_f16.hip", "conv16.hip", "act16.hip", "attn16.hip", "gemm_f32.hip", "groupnorm.hip", "misc.hip", "ddnm_step.hip", "fwht.hip", "backward.hip"]
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
    cmd = [_hipcc()] + FLAGS + [f'-DDDNM_BUILD_DIGEST="{dig}"'] + [os.path.join(CSRC, s) for s in SOURCES] + ["-o", tmp]
    if verbose:
        print(" ".join(cmd))
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stdout + r.stderr)
        raise RuntimeError("hipcc failed")
    if verbose and r.stderr:
        sys.stderr.write(r.stderr)
    os.replace(tmp, LIB)               # atomic: a concurrent loader never maps a half-written file
    with open(STAMP, "w") as f:
        f.write(dig)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
