import sys, time, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from oracle_lib import RefKernels, build_oracle
build_oracle(ref=True)
r = RefKernels(portable=True)
rng = np.random.default_rng(7)
x = (rng.standard_normal((1024 * 1024, 2)) * 0.05).astype(np.float32)
r.process(x[:16 * 1024])          # warm-up
t0 = time.time(); r.process(x); dt = time.time() - t0
print("reference kernels (oracle/_ref, fiber shim, 1 thread), C2 batch of 1024 spectra: %.2f s -> %.2f MS/s" % (dt, 1.048576 / dt))
