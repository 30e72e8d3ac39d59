#!/bin/bash
# On the GPU box: kernel durations (rocprofv3 --kernel-trace) of tools/imgconv_f16_probe.py per library variant.
OUT=gpurun_out/${TAG:-abr}; mkdir -p $OUT; export TMPDIR=/tmp
for lib in "$@"; do
  name=$(basename $lib .so)
  echo "== $lib" | tee -a $OUT/rocprof.txt
  MVK_LIB_PATH=$PWD/$lib timeout 300 rocprofv3 --kernel-trace -d $OUT/$name -o t -- python tools/imgconv_f16_probe.py > $OUT/$name.log 2>&1
  python tools/rocpd_summary.py $OUT/$name/t_results.db 2>/dev/null | grep -E "imgconv_kernel<.*, 2>|imgwgrad_kernel<.*, 2>" | cut -c1-120 | tee -a $OUT/rocprof.txt
  rm -rf $OUT/$name
done
