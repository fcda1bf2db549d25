#!/bin/bash
for f in 1 0; do echo "== RT_HIP_LDS=$f"; RT_HIP_LIB=dev RT_HIP_LDS=$f timeout -k 10 100 python -u tools/ablate.py cube lighthouse 2>&1 | grep -E "full" ; done
