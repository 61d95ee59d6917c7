cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
for v in "X=1" "RAFTX_REDUCE_PHASE2=1"; do
  env $v python bench.py --no-cpu-baseline --steps 30 --warmup 5 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$v', 'ms/step %.3f solve %.3f isolated %.3f' % (d['ms_per_step'], d['step_breakdown_ms']['solve_kernels_sum'], d['isolated_call']['ms_per_step']))"
done; done
