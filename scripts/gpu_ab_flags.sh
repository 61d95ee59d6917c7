#!/bin/bash
# A/B of compiler-flag variants of the library on ONE box: bench.py (with its built-in parity checks) per variant.
# Usage: bash scripts/gpu_ab_flags.sh   (variants: raft_amd/csrc/libraftx_hip_v_*.so)
cd ${GRAFT_REPO_ROOT:-$(pwd)}
for i in 1 2; do
for lib in raft_amd/csrc/libraftx_hip.so raft_amd/csrc/libraftx_hip_v_*.so; do
  RAFTX_HIP_LIB=$PWD/$lib timeout 300 python bench.py --no-cpu-baseline --steps 20 --warmup 5 --resident 2>&1 | tail -1 | python -c "
import sys, json
l = sys.stdin.read()
try:
    d = json.loads(l)
    print('%-28s ms/step %.3f solve %.3f resident %.3f  err %.1e' % ('$lib'.split('libraftx_hip')[-1], d['ms_per_step'], d['step_breakdown_ms']['solve_kernels_sum'], d['kernel_resident']['kernel_ms'], d['parity']['rao_max_rel_err_vs_reference']))
except Exception as e:
    print('$lib', 'FAILED', l[-300:])"
done; done
