#!/bin/bash
# Builds scripts/cotenant_check (see cotenant_check.cpp) against the in-tree libmit_hip.so.
set -e
cd "$(dirname "$0")/.."
python -m manga_image_translator_amd.build >/dev/null
hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/cotenant_check.cpp -o scripts/cotenant_check -Iinclude -Lmanga_image_translator_amd -lmit_hip \
      -Wl,-rpath,'$ORIGIN/../manga_image_translator_amd'
echo built scripts/cotenant_check
