#!/bin/bash
# same-box A/B of two builds of conv_wino_kernel (product = fragment reads scheduled before the MFMAs; tools/bin/libe4s_hip_norf.so = built with
# -DE4S_WINO_READS_FIRST=0) x the two wave tiles, alternating, 512 -> 512 @32^2 x16 (stats epilogue): ms per launch
for i in 1 2 3; do
  for lib in norf rf; do
    if [ $lib = norf ]; then export E4S_LIB_PATH=$GRAFT_REPO_ROOT/tools/bin/libe4s_hip_norf.so; else unset E4S_LIB_PATH; fi
    for wt in 0 1; do
      echo -n "$lib wt$wt: "; E4S_WINO_WT=$wt python tools/wino_ablate.py run 32 2>/dev/null | tr '\n' ' '; echo
    done
  done
done
