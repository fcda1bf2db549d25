#!/bin/bash
# usage: tools/ab_env.sh VAR "<values>" "<scenes>" [lib tag]
for rep in 1 2; do
for v in $2; do echo "== $1=$v (rep $rep)"; env $1=$v RT_HIP_LIB=$4 timeout -k 10 120 python -u tools/ablate.py $3 2>&1 | grep -E "full" ; done
done
