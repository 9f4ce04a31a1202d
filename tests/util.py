"""Helpers shared by the parity tests."""
import numpy as np

REL_TOL = 1e-4  # band magnitudes: 1e-4 relative (different f32 summation orders)


def click_track(n, seed, sr=44100, level=26000, noise=500, period=None, burst_hz=60.0):
    """A numpy click track for oracle-only tests (not bit-stable across numpy
    versions; anything committed as a fixture is stored as data)."""
    rng = np.random.default_rng(seed)
    x = rng.integers(-noise, noise + 1, n).astype(np.int32)
    period = period or int(sr * 60 / rng.integers(60, 181))
    t = np.arange(4096)
    burst = np.exp(-t / (0.030 * sr)) * np.sin(2 * np.pi * burst_hz * t / sr)
    for start in range(int(rng.integers(0, period)), n, period):
        m = min(4096, n - start)
        x[start:start + m] += (level * rng.uniform(0.6, 1.0) * burst[:m]).astype(np.int32)
    return np.clip(x, -32768, 32767).astype(np.int16)


def value_tolerance(ref_values, max_abs_sample, divisor):
    """|gpu - ref| <= REL_TOL * |ref| + floor.  The floor (REL_TOL * largest
    sample / divisor) only matters for bins whose magnitude is below one
    sample's worth of signal, where both the f32 FFT and the pruned DFT are
    dominated by rounding; it is five orders of magnitude below the decision
    threshold."""
    div = divisor if divisor else 1.0
    return REL_TOL * np.abs(ref_values) + REL_TOL * max(float(max_abs_sample), 1.0) / div


# How often the absolute floor was what let a value pass (printed at the end of a test session):
# the relative bound alone is the stated tolerance, the floor only covers bins near zero.
FLOOR_STATS = {"values": 0, "needed_floor": 0, "largest_ref_needing_floor": 0.0}


def values_close(got, ref_values, max_abs_sample, divisor):
    """Boolean array: |got - ref| within value_tolerance(); counts the values that pass only
    through the absolute floor."""
    got = np.asarray(got, dtype=np.float64)
    ref = np.asarray(ref_values, dtype=np.float64)
    err = np.abs(got - ref)
    ok = err <= value_tolerance(ref_values, max_abs_sample, divisor)
    floor_only = ok & (err > REL_TOL * np.abs(ref))
    FLOOR_STATS["values"] += int(ok.size)
    FLOOR_STATS["needed_floor"] += int(floor_only.sum())
    if floor_only.any():
        FLOOR_STATS["largest_ref_needing_floor"] = max(FLOOR_STATS["largest_ref_needing_floor"],
                                                       float(np.abs(ref[floor_only]).max()))
    return ok


def threshold_adjacent(ref_values, threshold):
    """Callbacks whose reference value is within REL_TOL of the threshold: the
    only ones whose decision may legitimately differ."""
    return np.abs(ref_values.astype(np.float64) - threshold) <= REL_TOL * threshold


def oracle_batch(oracle, kind, sr, samples2d, callback_len, params=None):
    vals, cands, beats = [], [], []
    for row in samples2d:
        v, c, b = oracle.run_stream(kind, sr, row, callback_len, params=params, full=True)
        vals.append(v); cands.append(c); beats.append(b)
    return np.stack(vals), np.stack(cands), beats
