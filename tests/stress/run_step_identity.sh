#!/bin/bash
# tests/stress/run_step_identity.sh [processes per config]: distinct result hashes over fresh processes (one line per distinct hash)
np=${1:-100}
here=$(dirname "$0")
for cfg in ${CFGS:-zinc counting sr25 mutag}; do
  for i in $(seq $np); do CFG=$cfg python3 $here/step_identity.py 2>/dev/null | tail -1; done | sort | uniq -c
done
