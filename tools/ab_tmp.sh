for i in 1 2 3; do
echo -n "HEAD   : "; python bench.py --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'])"
echo -n "frozen : "; python tools/bench_frozen_tmp.py --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'])"
done
