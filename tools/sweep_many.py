import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
import tests.test_gpu_random_sweep as sw
fails = 0
for seed in range(120, 1500):
    try:
        sw.test_random_configuration(seed)
    except Exception as e:
        fails += 1
        print("FAIL seed", seed, str(e)[:300].replace("\n", " "))
        if fails > 10: break
print("done, fails:", fails)
