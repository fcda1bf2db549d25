#!/bin/bash
# usage: tools/ab_variants.sh "<tags>" "<scenes>"   -- same box, same process order, full frames
for rep in 1 2; do
for t in $1; do echo "== RT_HIP_LIB=$t (rep $rep)"; RT_HIP_LIB=$t timeout -k 10 120 python -u tools/ablate.py $2 2>&1 | grep -E "full" ; done
done
