#!/usr/bin/env python3
"""GPU box: the randomized-parameter parity tests of tests/test_gpu_parity.py for a range of seeds, one line per seed, flushed
BEFORE the seed runs (so that a hang names its seed).  usage: fuzz_run.py orb|fast_orb|calls first last"""
import os, sys, time, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_parity as T
fn = {"orb": T.test_random_parameter_sets_match_oracle, "fast_orb": T.test_random_parameter_sets_fast_orb_match_oracle, "calls": T.test_random_call_sequences_match_oracle}[sys.argv[1]]
for seed in range(int(sys.argv[2]), int(sys.argv[3])):
    print("seed", seed, "...", end=" ", flush=True)
    t0 = time.time()
    try:
        fn(seed); print("ok %.1fs" % (time.time() - t0), flush=True)
    except AssertionError as e:
        print("FAIL %.1fs" % (time.time() - t0), str(e)[:300].replace("\n", " "), flush=True)
    except Exception as e:
        print("ERROR %.1fs" % (time.time() - t0), repr(e)[:300], flush=True)
