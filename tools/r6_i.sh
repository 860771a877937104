cd $GRAFT_REPO_ROOT
T=r6_i
timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -4 > gpurun_out/${T}_pytest.log
cat gpurun_out/${T}_pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 1500 python bench.py > gpurun_out/${T}_bench_default.json 2> gpurun_out/${T}_bench_default.err; python -c "
import json; d=json.loads(open('gpurun_out/${T}_bench_default.json').read().strip().splitlines()[-1]); print('default', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline_hbm']['frac'], d['cpu_baseline']['value'], d['n1_same_settings']['value'], bool(d['numerics']))"
