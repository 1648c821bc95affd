#!/bin/bash
# A/B of environment knobs on bench.py (run on the GPU box): one JSON line per run.  Usage: bash tools/knob_sweep.sh "KNOB=V ..." ...
run() { env $1 python bench.py --cpu-sample 0 --steps 20 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps({'knob': '$1', 'ms_per_step': round(r['ms_per_step'],3), 'ms_per_leapfrog': round(r['ms_per_leapfrog'],3), 'syrk_ms': round(r['roofline']['ms'],3), 'kbar_ms': round(r['roofline_leapfrog']['ms'],3), 'asm_ms': round(r['assembly']['ms'],3)}))"; }
REPS=${REPS:-2}
for rep in $(seq $REPS); do
  for k in "$@"; do run "$k"; done
done
