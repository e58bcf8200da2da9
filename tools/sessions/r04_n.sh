# round 4, session n: the whole GPU suite on the round's last sources + smoke
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_n; mkdir -p $O; cd $R
( time python -m pytest tests -m gpu -q ) > $O/pytest.log 2>&1; tail -n 6 $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -n 2
