python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "sign" 2>&1 | tail -15
for i in 1 2; do
echo -n "signs on : "; python bench.py --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], {k:v['avg_us'] for k,v in d['kernels'].items() if 'patch2' in k})"
echo -n "signs off: "; CGS_NO_SIGN_MASKS=1 python bench.py --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], {k:v['avg_us'] for k,v in d['kernels'].items() if 'patch2' in k})"
done
