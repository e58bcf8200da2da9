for i in 1 2 3; do
for V in "CGS_NONE=1" "CGS_NO_SIGN_MASKS=1"; do
echo -n "$V: "; env $V python bench.py --steps 40 --warmup 4 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['ms_per_step_median'])"
done; done
