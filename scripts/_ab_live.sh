#!/bin/bash
# same-box A/B of bench.py --roofline-only between the shipped library and copo_amd/lib/libcopo_hip_prof_base.so
for i in 1 2 3; do
  python bench.py --roofline-only 2>/dev/null | tail -1 | cut -c1-130
  python -c "
import os, sys, runpy
import copo_amd._libsel as S
S.PATH = os.path.abspath('copo_amd/lib/libcopo_hip_prof_base.so')
sys.argv = ['bench.py', '--roofline-only']
runpy.run_path('bench.py', run_name='__main__')" 2>/dev/null | tail -1 | cut -c1-130
done
