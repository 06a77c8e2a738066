#!/bin/bash
# A/B sweeps of one tuning variable on the default bench shape, interleaved repeats, one line per run:
#   tools/sweep_env.sh VAR "v1 v2 v3" [repeats] [extra bench args]
VAR=$1; VALS=$2; REP=${3:-2}; shift 3
for r in $(seq 1 $REP); do
  for v in $VALS; do
    env $VAR=$v python bench.py --no-cpu-baseline --other-configs 0 --kernel-roofline 0 --host-input 0 --steps 30 --warmup 5 "$@" 2>/dev/null \
      | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$VAR=$v', d['value'], d['ms_per_step'])"
  done
done
