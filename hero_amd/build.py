"""Build libhero_hip.so (gfx950) with hipcc.  No torch headers are needed: the library is a
plain C-ABI shared object (include/hero_hip.h) loaded through ctypes."""
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_obj")
LIB = os.path.join(HERE, "libhero_hip.so")
SOURCES = ["api.cpp", "comm.cpp", "gemm.hip", "gemm_ws.hip", "layernorm.hip", "attention.hip", "attention_mfma.hip", "attention_mfma_long.hip", "rows.hip", "head.hip", "loss.hip", "collate.hip", "probe.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
         "-ffp-contract=off"]


def _hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: cannot build libhero_hip.so")
    return exe


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    os.makedirs(OBJ, exist_ok=True)
    hipcc = _hipcc()
    headers = [os.path.join(CSRC, "common.h"), os.path.join(CSRC, "gemm_args.h"), os.path.join(CSRC, "gemm_ws_common.h"), os.path.join(CSRC, "attn_mfma.h"),
               os.path.join(os.path.dirname(HERE), "include", "hero_hip.h")]
    jobs = []
    objs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJ, os.path.splitext(src)[0] + ".o")
        objs.append(o)
        if force or _stale(o, [s] + headers):
            jobs.append([hipcc, "-x", "hip"] + FLAGS + ["-c", s, "-o", o])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n%s\n%s" % (" ".join(cmd), r.stderr[-4000:]))
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)

    for f in os.listdir(OBJ):                 # objects of sources that left SOURCES (they would still travel to the GPU box)
        if f.endswith(".o") and os.path.join(OBJ, f) not in objs:
            os.remove(os.path.join(OBJ, f))
    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(run, jobs))
    if jobs or force or _stale(LIB, objs):
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-ldl"])
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
