cd $GRAFT_REPO_ROOT
for split in "0.2,0.8" "0.25,0.75" "0.3,0.7" "0.35,0.65" "0.2,0.3,0.5" "0.15,0.35,0.5" "0.25,0.35,0.4"; do
  for inl in 0 1; do
  export RAFTX_SWEEP_SPLIT=$split RAFTX_SWEEP_GEN_INLINE=$inl
  python bench.py --no-cpu-baseline --steps 30 --warmup 3 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('split $split inline $inl value %.1f M ms %.3f gen %.2f solve %.2f' % (d['value']/1e6, d['ms_per_step'], d['step_breakdown_ms']['generation_kernels_sum'], d['step_breakdown_ms']['solve_kernels_sum']))"
  done
done
