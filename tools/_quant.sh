#!/bin/bash
for q in 0 1 2; do echo "RT_QUANT_TEST=$q"; RT_QUANT_TEST=$q timeout -k 10 120 python tools/count_traversal.py c2 c3 c5 || exit 1; done
