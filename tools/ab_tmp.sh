python -m pytest tests/test_gpu_ops.py tests/test_gpu_cyclegan.py tests/test_gpu_fuzz.py tests/test_gpu_fuzz_archs.py -x -q -m gpu 2>&1 | tail -5
for i in 1 2; do echo -n "cyclegan256: "; python bench.py --arch cyclegan256 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], {k:v['avg_us'] for k,v in d['kernels'].items() if 'dot' in k})"; done
