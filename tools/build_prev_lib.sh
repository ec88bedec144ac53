#!/bin/bash
# A/B aid: build the library of the LAST COMMIT beside the working tree's (e4s_amd/libe4s_prev.so; load it with E4S_LIB_PATH).
set -e
cd "$(dirname "$0")/.."
rm -rf e4s_amd/build/prev && mkdir -p e4s_amd/build/prev
git archive HEAD e4s_amd/csrc include | tar -x -C e4s_amd/build/prev
cd e4s_amd/build/prev
for f in e4s_amd/csrc/*.hip; do
  extra=""
  case "$(basename "$f")" in conv_region1w.hip|conv_wino1w.hip) extra="-fno-slp-vectorize";; esac      # e4s_amd/build.py PER_FILE_FLAGS
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $extra -Iinclude -Ie4s_amd/csrc -c "$f" -o "$(basename "$f").o" 2>/dev/null &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../libe4s_prev.so *.o
echo built e4s_amd/libe4s_prev.so from $(git rev-parse --short HEAD)
