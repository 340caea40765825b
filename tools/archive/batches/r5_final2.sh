#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5_final2; mkdir -p $O
( time timeout 3300 python -m pytest tests -m gpu -q 2>&1 | tail -6 ) 2>&1 | tee $O/pytest.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $O/smoke.txt
timeout 1200 python bench.py --steps 20 --warmup 3 > $O/bench.json 2> $O/bench.err
python -c "
import json; d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('traffic'), d['roofline'].get('traffic_from_this_build'), d['end_to_end']['first_ys_ms'], len(json.dumps(d)))
print({k:(v.get('ms_per_step'), v.get('frac_of_8TBps_at_8B_per_sample')) for k,v in d['variants'].items()})"
