#!/bin/bash
# Round-6 meta-pass A/B on one box: shipped library against A/B copies (copo_amd/lib/libcopo_hip_<tag>.so, `make variant`), alternating;
# then a kernel trace of the shipped one.   usage: scripts/ab_meta_r06.sh <tag> [<tag> ...]
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/ab_meta; mkdir -p $OUT
cd $ROOT
for rep in 1 2; do
  for tag in "$@" ship; do
    if [ $tag = ship ]; then lib=-; else lib=copo_amd/lib/libcopo_hip_$tag.so; fi
    echo -n "$tag: "; timeout 300 python scripts/ab_lib.py $lib 10 2>&1 | tail -n 1
  done
done | tee $OUT/ab.txt
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- python $ROOT/scripts/ab_lib.py - 6 > $OUT/trace.log 2>&1
python $ROOT/scripts/top_kernels.py $OUT/trace/trace_results.db 14 | tee $OUT/kernels.txt
python $ROOT/scripts/meta_timeline_db.py $OUT/trace/trace_results.db > $OUT/timeline.txt 2>&1
rm -rf $OUT/trace
