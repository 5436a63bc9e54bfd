#!/bin/bash
# Developer tool (GPU box): encoder forward time and per-kernel averages for several builds of the library
# (variants built with `make BUILD=build_x OUT=$ROOT/build_ab/libanorag_x.so EXTRA=-D... lib`, loaded through ANORAG_LIB).
#   tools/enc_ab.sh main nostag ...      ("main" = the in-tree library)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in "$@"; do
  if [ "$v" = main ]; then unset ANORAG_LIB; else export ANORAG_LIB=$R/build_ab/libanorag_$v.so; fi
  for rep in 1 2; do python3 $R/tools/enc_perf.py ${ENC_B:-256} ${ENC_L:-64} 2>/dev/null | sed "s/^/[$v] /"; done
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ab_$v -- python3 $R/tools/enc_perf.py ${ENC_B:-256} ${ENC_L:-64} > $R/gpurun_out/ab_$v.log 2>&1
  python3 $R/tools/kstats.py $R/gpurun_out/ab_$v | grep -E "gemm|layernorm|attention|embed|pool" | sed "s/^/[$v] /"
done
