"""The arithmetic fact behind k_blend's saturation skip, checked on the CPU in f32 exactly as the kernel computes a pixel
(w = T * a; T = T - w; C = fma(w, c, C) per channel; segments folded by C = fma(Tprefix, Ck, C), Tprefix *= Tk):
once T < 2^-27 * min(R, G, B) no later splat changes a bit of C, alpha = 1 - T is 1.0f, and that also holds through the
fold when the pixel finished inside a segment.  Independent of the GPU and of oracle.c."""
import numpy as np
import pytest

f32 = np.float32
K = f32(2.0 ** -27)


def fma32(a, b, c):
    # a*b is exact in float64 (24 + 24 bits); the sum is rounded once to 80-bit extended and once to f32 -- a double
    # rounding that can only differ from a true fma on ties of measure zero, and never in the regime under test
    return f32(np.longdouble(np.float64(a) * np.float64(b)) + np.longdouble(c))


def composite(weights, colours, skip, T0=f32(1.0)):
    """one pixel, one segment: returns (C[3], T, index at which the pixel finished or None)"""
    T = f32(T0)
    C = [f32(0.0)] * 3
    finished = None
    for i, (a, c) in enumerate(zip(weights, colours)):
        if skip and finished is not None:
            break
        w = f32(T * a)
        T = f32(T - w)
        C = [fma32(w, c[k], C[k]) for k in range(3)]
        if finished is None and (T < f32(K * min(C)) or T == 0):
            finished = i
    return C, T, finished


def fold(partials):
    C = [f32(0.0)] * 3
    T = f32(1.0)
    for Ck, Tk in partials:
        C = [fma32(T, Ck[k], C[k]) for k in range(3)]
        T = f32(T * Tk)
    return C, T


@pytest.mark.parametrize("seed", range(6))
def test_a_finished_pixel_never_changes_again(seed):
    rng = np.random.default_rng(seed)
    n = 900
    # weights like opacity * exp(-q): a mix of strong and faint fragments; colours in [0, 1] incl. exact 0 and 1
    a = (rng.uniform(0.0, 1.0, n) ** 2).astype(np.float32)
    a[rng.integers(0, n, 20)] = f32(1.0) if seed % 2 else f32(0.999)
    c = rng.integers(0, 256, (n, 3)).astype(np.float32) * f32(1.0 / 255.0)
    if seed == 3:
        c[:, 1] = 0.0          # a channel that stays exactly 0: the pixel may finish only at T == 0
    full_C, full_T, fin = composite(a, c, skip=False)
    skip_C, skip_T, fin2 = composite(a, c, skip=True)
    assert fin == fin2
    if fin is not None:
        assert fin < n - 1, "the sequence saturates well before its end"
    assert [float(x) for x in skip_C] == [float(x) for x in full_C]
    assert f32(1.0) - skip_T == f32(1.0) - full_T


@pytest.mark.parametrize("seed", range(4))
def test_finishing_inside_a_segment_survives_the_fold(seed):
    rng = np.random.default_rng(100 + seed)
    n, cut = 1400, (300, 1000)     # three segments; the middle one is long enough to finish on its own
    a = (rng.uniform(0.0, 1.0, n) ** 1.5).astype(np.float32)
    c = rng.integers(1, 256, (n, 3)).astype(np.float32) * f32(1.0 / 255.0)
    a[:cut[0]] *= f32(0.02)        # a faint first segment: the prefix transmittance stays large
    bounds = (0,) + cut + (n,)
    images = {}
    for skip in (False, True):
        parts = []
        for s0, s1 in zip(bounds[:-1], bounds[1:]):
            Ck, Tk, fin = composite(a[s0:s1], c[s0:s1], skip)
            if s0 == cut[0]:
                assert fin is not None and fin < s1 - s0 - 1, "the middle segment finishes before its end"
            parts.append((Ck, Tk))
        C, T = fold(parts)
        images[skip] = ([float(x) for x in C], float(f32(1.0) - T))
    assert images[True] == images[False]
