import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
import conftest  # noqa
import oracle as oracle_mod
import test_gpu_dispatch_fuzz as f
import pragma_dsp_amd as pdsp
bad = 0
for seed in range(100, 140):
    try:
        f.test_transform_dispatch_fuzz.__wrapped__(oracle_mod, seed) if hasattr(f.test_transform_dispatch_fuzz, "__wrapped__") else f.test_transform_dispatch_fuzz(oracle_mod, seed)
        f.test_spectrum_dispatch_fuzz(oracle_mod, seed)
        f.test_fused_peaks_fuzz(oracle_mod, seed)
        if seed % 4 == 0:
            f.test_host_dropin_spectrum_fuzz(pdsp, oracle_mod, seed)
    except AssertionError as e:
        bad += 1
        print("seed", seed, "FAILED:", str(e)[:300])
print("extra fuzz done, failures:", bad)
