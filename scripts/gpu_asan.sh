#!/bin/bash
# SURVEY.md section 5 (sanitizers) for the PRODUCT library, not just the checker:
#   host side of libraftx_hip.so under AddressSanitizer + UndefinedBehaviorSanitizer (clang's shared runtime preloaded into
#   python): the streamed / staged / cancelled crossings, the chunked crossings, the communicator tests, a 300-step bench soak;
#   device side (optional, RAFTX_DEVICE_ASAN=<path to an xnack+ -fsanitize=address build>): raftx_build_designs + a 64-design
#   solve under HSA_XNACK=1.
# Build (in the build container, the .so travels):  bash scripts/gpu_asan.sh build
# Run on the GPU box:                                bash scripts/gpu_asan.sh run <tag>
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
LIB=$R/raft_amd/csrc/libraftx_hip_asan.so
DLIB=$R/raft_amd/csrc/libraftx_hip_dasan.so
if [ "${1:-run}" = "build" ]; then
  hipcc --offload-arch=gfx950 -O1 -g1 -std=c++17 -shared -fPIC -fno-gpu-rdc -fsanitize=address,undefined -fno-sanitize=function,vptr \
        -shared-libsan -fno-omit-frame-pointer -Wno-option-ignored -o $LIB $R/raft_amd/csrc/raftx_hip.hip || exit 1
  echo built $LIB; exit 0
fi
TAG=${2:-r04_asan}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
# Runtime: clang's own shared ASan runtime (ROCm build) intercepts hsa_amd_memory_pool_allocate for device-side ASan and aborts
# at HIP start-up on this pool's driver ("out of memory", with HSA_XNACK=0 and 1 alike: profiles/r04_asan.txt), which also rules
# out an xnack+ device-ASan build here.  The host instrumentation is runtime-ABI v8, the same as GCC 11's libasan / libubsan:
# those are preloaded instead, under the DT_NEEDED name the library was linked against.
mkdir -p /tmp/raftx_san && ln -sf $(readlink -f $(gcc -print-file-name=libasan.so)) /tmp/raftx_san/libclang_rt.asan-x86_64.so
RT="$(readlink -f $(gcc -print-file-name=libasan.so)) $(readlink -f $(gcc -print-file-name=libubsan.so))"
export LD_LIBRARY_PATH=/tmp/raftx_san:${LD_LIBRARY_PATH:-}
export ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0:abort_on_error=0:halt_on_error=0:detect_odr_violation=0:log_path=$OUT/asan
export UBSAN_OPTIONS=print_stacktrace=1:log_path=$OUT/ubsan
export RAFTX_HIP_LIB=$LIB
{
echo "== host ASan + UBSan: $LIB  (runtime $RT)"
echo "-- the runtime is live in the python process and the library's accesses are instrumented:"
( LD_PRELOAD="$RT" ASAN_OPTIONS=$ASAN_OPTIONS:verbosity=1:log_path=stderr python -c "import ctypes, os; ctypes.CDLL(os.environ['RAFTX_HIP_LIB'])" 2>&1 | grep -m1 "Init done" )
nm -D $LIB | grep -c " U __asan_report_\| U __ubsan_handle_" | sed 's/^/   instrumentation call sites: distinct __asan_report_* \/ __ubsan_handle_* imports = /'
( LD_PRELOAD="$RT" timeout 1200 python -m pytest tests/test_geometry.py tests/test_hip_comm.py tests/test_hip_parity.py tests/test_flexible.py -m gpu -x -q \
    -k "streamed or crossing or comm or featured or ragged or singular or dense or flex or variant or farm or strip_exports or exchange or mooring" 2>&1 | tail -6 )
echo "== 300-step soak of the streamed crossing (bench.py --no-cpu-baseline --no-extra-legs --steps 300)"
( LD_PRELOAD="$RT" timeout 900 python bench.py --no-cpu-baseline --no-extra-legs --steps 300 --warmup 3 2>&1 | tail -1 | cut -c1-260 )
echo "== 40 steps with the responses downloaded (four batches in flight)"
( LD_PRELOAD="$RT" timeout 900 python bench.py --no-cpu-baseline --no-extra-legs --xi-out --steps 40 --warmup 3 2>&1 | tail -1 | cut -c1-260 )
echo "== isolated crossings with the responses out (fused launches in slabs, each with its download)"
( LD_PRELOAD="$RT" ISO_CALLS=8 timeout 600 python scripts/iso_xi.py 2>&1 | tail -2 )
echo "== sanitizer reports"
ls $OUT | grep -c "asan\.\|ubsan\." | sed 's/^/report files: /'
cat $OUT/asan.* $OUT/ubsan.* 2>/dev/null | head -80
if [ -f "$DLIB" ]; then
  echo "== device ASan (xnack+): $DLIB"
  ( HSA_XNACK=1 RAFTX_HIP_LIB=$DLIB LD_PRELOAD="$RT" timeout 600 python - <<'PY' 2>&1 | tail -12
import numpy as np, json
from raft_amd import backend, snapshot, geometry as G
from tests.util import volturnus_sweep
fx = snapshot.load_fixture("c3_variants.npz"); fg = snapshot.load_fixture("geom_units.npz")
ctx = backend.hip_library().context(0)
n = 64
D = volturnus_sweep(json.loads(fg["c3_base_json"]), np.asarray(fx["scales"])[:n]).tables()
u0 = [u for u in fg["units"] if u["name"] == "C3-variant-0"][0]
M = np.repeat((np.asarray(u0["M_struc"]) - np.asarray(u0["M_struc_bare"]))[None], n, 0)
C = np.repeat((np.asarray(u0["C_struc"]) - np.asarray(u0["C_struc_bare"]) + np.diag([7e4, 7e4, 0, 0, 0, 1e8]))[None], n, 0)
off = ctx.build_designs(D.member_off, D.members, D.station_off, D.stations, M, np.repeat(np.asarray(fx["B0"])[:1], n, 0), C, 200, cap_off=D.cap_off, caps=D.caps, add_mask=7)
ctx.upload_cases(fx["w"], fx["k"], float(fx["depth"]), 1025.0, 9.81, np.asarray(fx["zeta"])[None], np.asarray(fx["beta"])[None])
ctx.solve_dynamics_device(int(fx["nIter"]), 0.01, float(fx["XiStart"]))
r = ctx.fetch_results(want_Xi=True)
err = max(np.abs(r["Xi"][j, 0, 0] - np.asarray(s["Xi"])[0]).max() / np.abs(np.asarray(s["Xi"])[0]).max() for j, s in enumerate(fx["solved"][:n]))
print("device-ASan build: 64 designs generated + solved, max rel err vs the live reference %.2e, niter ok %s" % (err, all(int(r["niter"][j, 0]) == int(s["units"][0]["niter"]) for j, s in enumerate(fx["solved"][:n]))))
PY
  )
fi
} 2>&1 | tee $OUT/report.txt
