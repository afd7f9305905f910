#!/bin/bash
# Builds scripts/split_check (stand-alone split-bf16 tile check; see split_check.cpp) against the in-tree libmit_hip.so.
set -e
cd "$(dirname "$0")/.."
python -m manga_image_translator_amd.build >/dev/null
hipcc --offload-arch=gfx950 -O2 -std=c++17 scripts/split_check.cpp -o scripts/split_check -Iinclude -Lmanga_image_translator_amd -lmit_hip \
      -Wl,-rpath,'$ORIGIN/../manga_image_translator_amd'
echo built scripts/split_check
