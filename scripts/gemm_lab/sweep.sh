#!/bin/bash
cd "$(dirname "$0")"
for b in lab_d*; do
  echo "== $b"
  timeout 200 ./$b ${2:-100} 39129 512 $1 2>&1 | grep -v "occupancy\|bad\|rows in\|cols in\|tiles:\|vs fp64\|ranks"
done
