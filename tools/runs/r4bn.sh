set -e
O=gpurun_out/r4bn; mkdir -p $O
for i in 1 2; do python tools/ab_time.py c5 6 2>&1 | grep -v amdgpu >> $O/t.txt; done
cat $O/t.txt
python -m pytest tests -m gpu -x -q -k "large or grid or config5 or spheres or rays" > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
tail -2 $O/tests.log
