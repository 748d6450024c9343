set -e
mkdir -p gpurun_out/r4d
export RPT_DISPATCH_TIMELINE=1
for o in 2 1; do
for c in c2 share c5; do
  echo "== $c order mode $o" >> gpurun_out/r4d/timeline.txt
  RPT_DISPATCH_ORDER=$o python tools/dispatch_timeline.py $c >> gpurun_out/r4d/timeline.txt 2>&1
done
done
grep -v amdgpu.ids gpurun_out/r4d/timeline.txt
