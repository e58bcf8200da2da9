python -m pytest tests/test_gpu_ops.py -x -q -m gpu 2>&1 | tail -8
python -m pytest tests/test_gpu_refine.py tests/test_gpu_fuzz.py tests/test_gpu_fullsize.py -x -q -m gpu -k "not dcgan64 and not cyclegan" 2>&1 | tail -3
for i in 1 2; do echo -n "mnist: "; python bench.py --arch mnist --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], {k:v['avg_us'] for k,v in d['kernels'].items() if 'taps' in k or 'rows' in k})"; done
