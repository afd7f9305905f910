cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r07a; mkdir -p $OUT
python - > $OUT/probe_imports.log 2>&1 <<'PY'
import importlib
for m in ("cv2","shapely","pyclipper","torchvision","pydensecrf","PIL","skimage","kornia"):
    try:
        mod=importlib.import_module(m); print(m,"OK",getattr(mod,"__version__","?"))
    except Exception as e:
        print(m,"MISSING",type(e).__name__,str(e)[:80])
import subprocess
print(subprocess.run("pip download opencv-python-headless --no-deps -d /tmp/x 2>&1 | tail -2", shell=True, capture_output=True, text=True, timeout=60).stdout)
print(subprocess.run("rocm-smi --showtopo 2>&1 | head -20; nproc", shell=True, capture_output=True, text=True).stdout)
PY
cat $OUT/probe_imports.log
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log; tail -5 $OUT/pytest.log
timeout 900 python bench.py --prof-dump $OUT/layers.csv > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
head -c 600 $OUT/bench.json; echo; tail -3 $OUT/bench.err
