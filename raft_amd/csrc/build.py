"""Builds raft_amd/csrc/libraftx_hip.so for gfx950 with hipcc (cross-compiles
without a GPU).  In-tree output so the .so travels to the GPU box."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "raftx_hip.hip")
OUT = os.path.join(HERE, "libraftx_hip.so")
# -disable-machine-licm: the register-starved kernels of this library (the fused fixed point sits at 256 VGPRs with ~120 SGPRs
# spilled to lanes) lose more to loop-invariant literals hoisted into registers -- and then spilled -- than they gain:
# with it off the fused kernel needs 247 VGPRs / 80 spilled SGPRs, 16 of the 41 specialisations that had scratch lose it
# (k_linearize at the drop-in's shape among them), the fused kernels run 1-2 % faster and the QTF kernels 1 % slower
# (same-box A/B of the whole bench: profiles/r06_experiments/machine_licm_off_whole_bench_ab.txt).
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-fno-gpu-rdc", "-mllvm", "-disable-machine-licm"]


def build(force=False, verbose=False, extra=()):
    deps = [SRC, os.path.join(HERE, "..", "..", "include", "raftx.h"), os.path.abspath(__file__)]
    deps += [os.path.join(HERE, f) for f in os.listdir(HERE) if f.endswith(".h")]
    if (not force) and os.path.exists(OUT) and all(os.path.getmtime(OUT) >= os.path.getmtime(d) for d in deps):
        return OUT
    hipcc = os.environ.get("HIPCC", "hipcc")
    cmd = [hipcc] + FLAGS + list(extra) + ["-o", OUT, SRC]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return OUT


def build_timing(verbose=False):
    """Instrumented tuning build (per-phase cycle counters of the fused kernel, read by scripts/phase_timing.py, and of
    k_geom_design, printed to stderr when the ctx is destroyed): libraftx_hip_timing.so.  Never loaded by the product;
    select it with RAFTX_HIP_LIB=<path>."""
    out = os.path.join(HERE, "libraftx_hip_timing.so")
    cmd = [os.environ.get("HIPCC", "hipcc")] + FLAGS + ["-DRAFTX_PHASE_TIMING", "-DGEOM_PHASE_TIMING", "-o", out, SRC]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return out


if __name__ == "__main__":
    if "--timing" in sys.argv:
        print(build_timing(verbose=True))
        sys.exit(0)
    print(build(force="--force" in sys.argv, verbose=True,
                extra=["-Rpass-analysis=kernel-resource-usage"] if "--usage" in sys.argv else []))
