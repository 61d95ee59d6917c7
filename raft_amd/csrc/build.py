"""Builds raft_amd/csrc/libraftx_hip.so for gfx950 with hipcc (cross-compiles
without a GPU).  In-tree output so the .so travels to the GPU box."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "raftx_hip.hip")
OUT = os.path.join(HERE, "libraftx_hip.so")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-fno-gpu-rdc"]


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


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True,
                extra=["-Rpass-analysis=kernel-resource-usage"] if "--usage" in sys.argv else []))
