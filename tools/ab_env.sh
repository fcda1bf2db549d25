#!/bin/bash
# usage: tools/ab_env.sh "<VAR=a VAR=b ...>" "<scenes>"  -- same box, the development build of the library (the switches exist only there), one environment setting per run
for rep in 1 2; do
for e in $1; do echo "== $e (rep $rep)"; env RT_HIP_LIB=dev $e timeout -k 10 120 python -u tools/ablate.py $2 2>&1 | grep -E "full" ; done
done
