#!/bin/bash
# tests/stress/run_fwd4.sh [processes per shape]: the chunked ring forward (default library), fresh processes, repeat-and-compare
np=${1:-32}
here=$(dirname "$0")
for cfg in "6 32 13" "6 48 13" "4 48 26"; do
  set -- $cfg
  tot=0; fails=0
  for i in $(seq $np); do
    k=$(DS=$1 DF=$2 DEG=$3 DN=40000 REPS=16 python3 $here/fwd4_repeat.py 2>&1 | tail -1 | awk '{print $NF}')
    tot=$((tot+16)); fails=$((fails+k))
  done
  echo "S=$1 Fin=$2 deg=$3: failing launches $fails of $tot"
done
