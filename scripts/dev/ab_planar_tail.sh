cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r10j
timeout 600 python -m pytest tests/test_lama_gpu.py -m gpu -q -k "planar_tail or page_parity" 2>&1 | tail -4
for i in 1 2; do for m in 4 16; do echo "--- MIT_LAMA_PLANAR_TAIL=$m"; B=16 MIT_LAMA_PLANAR_TAIL=$m python scripts/bench_lama.py 2>&1 | tail -1; done; done | tee gpurun_out/r10j/ab_planar_tail.log
for m in 4 16; do MIT_LAMA_PLANAR_TAIL=$m B=16 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r10j/prof$m -o t -- python scripts/bench_lama.py > /dev/null 2>&1; f=$(find gpurun_out/r10j/prof$m -name "*kernel_stats.csv" | head -1); echo "tail $m:"; grep -i "small_cout\|<128, 64, 16, 2, 2, 3, 6, 385>" $f | cut -d, -f1-4 | cut -c1-160; cp $f gpurun_out/r10j/kernel_stats_tail$m.csv; rm -rf gpurun_out/r10j/prof$m; done
