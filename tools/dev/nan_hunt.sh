#!/bin/bash
# Runs tools/dev/nan_hunt.py in fresh processes, one after the other; stops at the first timeout.
# usage: nan_hunt.sh <runs> <logfile> [flags...]
runs=$1; log=$2; shift 2
mkdir -p "$(dirname "$log")"
ok=0; badn=0
for i in $(seq 1 "$runs"); do
  timeout -k 10 180 python tools/dev/nan_hunt.py --tag "run$i[$*]" "$@" >> "$log" 2>&1
  rc=$?
  if [ $rc -eq 0 ]; then ok=$((ok+1)); elif [ $rc -eq 1 ]; then badn=$((badn+1)); else echo "run $i rc $rc: stopping" >> "$log"; break; fi
done
echo "SUMMARY [$*] ok=$ok bad=$badn" | tee -a "$log"
