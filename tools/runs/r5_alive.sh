O=gpurun_out/r5_alive; mkdir -p $O
for c in c2 c4 c5; do spp=256; [ $c = c4 ] && spp=64; [ $c = c5 ] && spp=32
  timeout -k 10 200 python tools/block_profile.py $spp $c 2>&1 | grep -v amdgpu.ids > $O/bp_$c.txt || exit 1; done
timeout -k 10 100 python tools/block_profile.py 64 c2 2>&1 | grep -v amdgpu.ids > $O/bp_c2_64.txt
grep -h "profiled\|lanes with" $O/*.txt
