# usage: tools/builder_ab.sh  -- traversal counters and throughput for the three BVH builders
for b in 1 3 2; do export RT_BVH_BUILDER=$b; echo "builder $b"; timeout -k 10 200 python -u tools/count_traversal.py c2 c5 2>&1 | grep "^c" | cut -c1-150; timeout -k 10 200 python -u bench.py --no-cpu --steps 30 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c2 bench', round(d['value']), 'traverse-only', round(d['roofline']['traverse_Mrays_s']), d['build'])"; timeout -k 10 200 python -u bench.py --no-cpu --workload c5 --steps 10 --warmup 2 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c5 bench', round(d['value']))"; done
