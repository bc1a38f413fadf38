"""Search for camera-frame points whose pinhole projection u = (float)((fx*X + cx*Z) / Z) has its fp64
quotient within a few ulp of a float rounding boundary: the cases where a reciprocal-multiply
estimate of the quotient and the exact division can round to different floats.  Plain numpy, seeded;
writes tests/golden/projection_canaries.npz (camera-frame x, y, z as float32)."""
import numpy as np

FX = FY = 320.0
CX, CY = 320.0, 240.0
rng = np.random.default_rng(20260104)
keep = []
for it in range(400):
    n = 2_000_000
    X = rng.uniform(-8.0, 8.0, n).astype(np.float32)
    Y = rng.uniform(-2.0, 2.0, n).astype(np.float32)
    Z = rng.uniform(4.0, 40.0, n).astype(np.float32)
    Xd, Yd, Zd = X.astype(np.float64), Y.astype(np.float64), Z.astype(np.float64)
    for num in (((FX * Xd + 0.0 * Yd) + CX * Zd), ((0.0 * Xd + FY * Yd) + CY * Zd)):
        q = num / Zd
        low = q.view(np.uint64) & np.uint64(0x1FFFFFFF)
        d = np.abs(low.astype(np.int64) - 0x10000000)
        sel = np.flatnonzero(d <= 6)
        for i in sel:
            keep.append((X[i], Y[i], Z[i]))
    if len(keep) >= 64:
        break
pts = np.array(keep[:64], dtype=np.float32)
print(len(keep), "canaries after", it + 1, "rounds")
np.savez("tests/golden/projection_canaries.npz", x=pts[:, 0], y=pts[:, 1], z=pts[:, 2])
