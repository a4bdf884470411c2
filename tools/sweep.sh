#!/bin/bash
# usage: tools/sweep.sh ENVVAR v1 v2 ...   -> prints fps per value
var=$1; shift
for v in "$@"; do
  env $var=$v python bench.py --no-cpu-baseline 2>/dev/null | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$var=$v', d['value'], d['phase_ms_per_step'])"
done
