cd "$(dirname "$0")/.."
for r in 1 2; do for v in 2 4 8 16; do for p in 12 1; do
ZG_MSM_HEAVY=$v python3 bench.py --provers $p --steps $([ $p = 1 ] && echo 12 || echo 10) --warmup 3 --tail-only-headline --no-kernel-events 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
print('ZG_MSM_HEAVY=$v provers $p round $r: %.4f ms/proof' % d['ms_per_proof'])"
done; done; done
