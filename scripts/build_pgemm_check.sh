#!/bin/bash
# Builds scripts/pgemm_check (stand-alone check of mit_pgemm; see pgemm_check.cpp) against the in-tree libmit_hip.so.
set -e
cd "$(dirname "$0")/.."
python -m manga_image_translator_amd.build >/dev/null
hipcc --offload-arch=gfx950 -O2 -std=c++17 scripts/pgemm_check.cpp -o scripts/pgemm_check -Iinclude -Lmanga_image_translator_amd -lmit_hip \
      -Wl,-rpath,'$ORIGIN/../manga_image_translator_amd'
echo built scripts/pgemm_check
