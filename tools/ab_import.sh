#!/bin/bash
# layout import A/B: row segments through LDS-DMA (default) against the column-walking kernel (MPDATA_LAYOUT_NOROWS=1)
for pass in 1 2 3; do
  echo "pass $pass rows   : $(python3 tools/uw_bench.py --no-plan --no-uw --sets 6 | grep 'import\|export' | tr '\n' '|')"
  echo "pass $pass columns: $(MPDATA_LAYOUT_NOROWS=1 python3 tools/uw_bench.py --no-plan --no-uw --sets 6 | grep 'import\|export' | tr '\n' '|')"
done
