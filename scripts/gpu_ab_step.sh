#!/bin/bash
# same-box A/B of library variants on the streamed step (K = 40, three interleaved repeats): bash scripts/gpu_ab_step.sh v1.so v2.so ...
cd ${GRAFT_REPO_ROOT:-$(pwd)}
one() { python bench.py --no-cpu-baseline --no-extra-legs --steps 40 --warmup 5 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("%.3f ms/step (kernel %.3f, gen %.3f)" % (d["ms_per_step"], d["roofline"]["kernel_ms_per_step"], d["step_breakdown_ms"]["generation_kernels_sum"]))'; }
for rep in 1 2 3; do
  echo "head: $(one)"
  for v in "$@"; do echo "$(basename $v): $(RAFTX_HIP_LIB=$PWD/$v one)"; done
done
