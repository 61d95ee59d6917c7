#!/bin/bash
# HIP API + kernel + copy timeline of a few streamed steps (who waits for whom).  Usage: bash scripts/gpu_trace_api.sh <tag> [bench args]
set -u
TAG=${1:-apitrace}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --hip-runtime-trace --kernel-trace --memory-copy-trace --output-format csv -d $OUT/trace -o bench -- python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline "$@" > $OUT/trace.log 2>&1
cd $R
ls -la $OUT/trace | head
python3 - $OUT/trace <<'PY'
import csv, sys, os
d = sys.argv[1]
ev = []
for r in csv.DictReader(open(os.path.join(d, 'bench_kernel_trace.csv'))):
    ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'GPU  ' + r['Kernel_Name'][:40]))
for r in csv.DictReader(open(os.path.join(d, 'bench_memory_copy_trace.csv'))):
    ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'COPY ' + r['Direction'][12:]))
api = os.path.join(d, 'bench_hip_api_trace.csv')
for r in csv.DictReader(open(api)):
    n = r['Function']
    if n in ('hipLaunchKernel', 'hipEventSynchronize', 'hipStreamSynchronize', 'hipEventQuery', 'hipMemcpyAsync', 'hipStreamWaitEvent', 'hipModuleLaunchKernel', 'hipExtModuleLaunchKernel', 'hipMalloc', 'hipHostMalloc', 'hipFree'):
        s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        if n == 'hipEventQuery' and e - s < 50000: continue
        ev.append((s, e, 'API  ' + n))
ev.sort()
sol = [e for e in ev if 'k_solve_dynamics' in e[2]]
w0 = sol[-16][0] - 300000 if len(sol) >= 16 else ev[0][0]
out = []
for e in ev:
    if e[0] < w0 or e[0] > w0 + 14_000_000: continue
    dur = (e[1] - e[0]) / 1e6
    if e[2].startswith('API') and dur < 0.03 and 'LaunchKernel' not in e[2]: continue
    out.append("%9.3f +%7.3f %s" % ((e[0] - w0) / 1e6, dur, e[2]))
open(os.path.join(os.path.dirname(d), 'timeline.txt'), 'w').write('\n'.join(out))
print('\n'.join(out[:140]))
PY
rm -f $OUT/trace/bench_hip_api_trace.csv
