# round 6: the parity suite's three fuzzers on 600 more seeds (1 800 random small / SDF / large scenes against the oracle, bit for bit) through
# the round's new kernel paths: the SDF kernel's second room, the material table by class (random small scenes of 5-12 primitives take it
# when their accepted sets fall into <= 16 classes), ragged waves
O=gpurun_out/r6_soak; mkdir -p $O
timeout -k 10 1000 python tools/fuzz_more.py ${1:-7000} ${2:-600} 2>&1 | grep -v amdgpu.ids | tee $O/fuzz_more.txt | tail -4
