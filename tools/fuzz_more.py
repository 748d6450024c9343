"""More seeds through the parity suite's fuzzers (tests/test_gpu_parity.py: random small / SDF / large scenes against the oracle, bit for
bit) than the suite itself runs:   python tools/fuzz_more.py <first seed> <count>      (on the GPU box)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import conftest
import test_gpu_parity as T

first, count = int(sys.argv[1]), int(sys.argv[2])
rpt = conftest.load_package()
conftest._build_oracle()
import oracle_lib
oracle = oracle_lib.Oracle("liboracle.so")
bad = 0
for seed in range(first, first + count):
    for fn in (T.test_random_small_scenes_match_oracle, T.test_random_sdf_scenes_match_oracle, T.test_random_large_scenes_match_oracle_in_both_forms):
        try:
            fn(rpt, oracle, seed)
        except AssertionError as e:
            bad += 1
            print("MISMATCH", fn.__name__, seed, str(e)[:300], flush=True)
    if (seed - first) % 20 == 19:
        print("seeds %d..%d done, %d mismatches" % (first, seed, bad), flush=True)
print("FUZZ: %d seeds x 3 scene classes, %d mismatches" % (count, bad))
sys.exit(1 if bad else 0)
