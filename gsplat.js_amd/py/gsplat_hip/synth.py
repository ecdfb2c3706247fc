"""Synthetic .splat scenes for the bench and the parity tests.

The reference ships no data files (its loaders fetch .splat/.ply from the
network, README.md:56), so every workload here is a seeded synthetic scene in
the reference's own 32-byte .splat row format (src/core/Scene.ts:9,126-148):

    [0..11] position f32x3 | [12..23] scale f32x3 | [24..27] r,g,b,a u8
    [28..31] rotation u8x4 = (w,x,y,z), decoded as (b-128)/128

The PRNG is mulberry32 used as a counter generator (call k has state
seed + k*0x6D2B79F5), 24 draws per splat, so scenes are reproducible and
vectorise in numpy.
"""
import numpy as np

ROW = 32  # Scene.RowLength, src/core/Scene.ts:9
DRAWS = 24

# name -> (seed, N, width, height, sigma, scale_lo, scale_hi, fx)   BASELINE.json configs
CONFIGS = {
    "C1": dict(seed=1, n=10_000, width=640, height=480, sigma=1.5, s_lo=0.004, s_hi=0.06, fx=1132.0),
    "C2": dict(seed=2, n=300_000, width=1920, height=1080, sigma=1.0, s_lo=0.003, s_hi=0.04, fx=1132.0),
    "C3": dict(seed=3, n=1_000_000, width=1920, height=1080, sigma=1.5, s_lo=0.004, s_hi=0.06, fx=1132.0),
    "C4": dict(seed=4, n=5_000_000, width=3840, height=2160, sigma=1.5, s_lo=0.004, s_hi=0.06, fx=2264.0),
}


def mulberry32(seed, first_call, count):
    """Outputs of calls first_call .. first_call+count-1 (1-based) as u32."""
    with np.errstate(over="ignore"):
        k = np.arange(first_call, first_call + count, dtype=np.uint64)
        t = ((np.uint64(seed) + k * np.uint64(0x6D2B79F5)) & np.uint64(0xFFFFFFFF)).astype(np.uint32)
        t = (t ^ (t >> np.uint32(15))) * (t | np.uint32(1))
        t = t ^ (t + (t ^ (t >> np.uint32(7))) * (t | np.uint32(61)))
        return t ^ (t >> np.uint32(14))


def synth_rows(n, seed, sigma=1.5, s_lo=0.004, s_hi=0.06, chunk=1 << 20):
    """n rows of .splat bytes as a uint8 array of length 32*n."""
    out = np.empty((n, ROW), dtype=np.uint8)
    for start in range(0, n, chunk):
        m = min(chunk, n - start)
        u = mulberry32(seed, start * DRAWS + 1, m * DRAWS).astype(np.float64) / 4294967296.0
        u = u.reshape(m, DRAWS)

        def normal(a, b):
            return np.sqrt(-2.0 * np.log(1.0 - u[:, a])) * np.cos(2.0 * np.pi * u[:, b])

        pos = np.stack([normal(0, 1), normal(2, 3), normal(4, 5)], axis=1) * sigma
        pos = np.clip(pos, -6.0, 6.0).astype(np.float32)
        scale = np.exp(np.log(s_lo) + u[:, 6:9] * (np.log(s_hi) - np.log(s_lo))).astype(np.float32)
        rgb = np.floor(256.0 * u[:, 9:12]).astype(np.uint8)
        alpha = (32 + np.floor(224.0 * u[:, 12])).astype(np.uint8)
        q = np.stack([normal(13, 14), normal(15, 16), normal(17, 18), normal(19, 20)], axis=1)
        q /= np.maximum(np.linalg.norm(q, axis=1, keepdims=True), 1e-12)
        rot = np.clip(np.round(q * 128.0 + 128.0), 0, 255).astype(np.uint8)

        blk = out[start:start + m]
        blk[:, 0:12] = pos.view(np.uint8).reshape(m, 12)
        blk[:, 12:24] = scale.view(np.uint8).reshape(m, 12)
        blk[:, 24:27] = rgb
        blk[:, 27] = alpha
        blk[:, 28:32] = rot
    return out.reshape(-1)


def config_rows(name):
    c = CONFIGS[name]
    return synth_rows(c["n"], c["seed"], c["sigma"], c["s_lo"], c["s_hi"])


# ---------------------------------------------------------------------------
# Scenes that are NOT one centred isotropic blob (scripts/policy_check.py: the per-bin work-item policy must hold where a
# frame mixes bins that saturate with bins that do not, as real captures -- a dense object in front of a sparse background
# -- do).  Built from synth_rows by editing the rows' position / alpha fields; same seeded generator underneath.
# ---------------------------------------------------------------------------
def _edit(rows, offset=None, alpha_min=None, on_sphere=None):
    r = np.array(rows, dtype=np.uint8).reshape(-1, ROW)
    pos = r[:, 0:12].copy().view(np.float32).reshape(-1, 3)
    if on_sphere is not None:        # points on a sphere of that radius (direction of the blob's sample)
        ln = np.maximum(np.linalg.norm(pos.astype(np.float64), axis=1, keepdims=True), 1e-9)
        pos = (pos / ln * on_sphere).astype(np.float32)
    if offset is not None:
        pos = (pos + np.asarray(offset, dtype=np.float32)).astype(np.float32)
    r[:, 0:12] = pos.view(np.uint8).reshape(-1, 12)
    if alpha_min is not None:
        r[:, 27] = np.maximum(r[:, 27], alpha_min)
    return r.reshape(-1)


def cluster_in_halo(n_cluster=300_000, n_halo=300_000, seed=21):
    """A tight cluster (sigma 0.4) inside a sparse halo (sigma 3): dense bins in the middle of the screen, thin ones around."""
    return np.concatenate([synth_rows(n_cluster, seed, 0.4, 0.004, 0.06), synth_rows(n_halo, seed + 1, 3.0, 0.004, 0.06)])


def two_clusters(n_each=400_000, seed=23):
    """Two clusters at different depths and screen positions (sigma 0.6 each, 3 units apart along x and z)."""
    a = _edit(synth_rows(n_each, seed, 0.6, 0.004, 0.06), offset=(-1.5, 0.0, -1.5))
    b = _edit(synth_rows(n_each, seed + 1, 0.6, 0.004, 0.06), offset=(1.5, 0.3, 1.5))
    return np.concatenate([a, b])


def opaque_shell(n=600_000, seed=25, radius=2.5):
    """A thin opaque shell: points on a sphere, alpha >= 240 -- every pixel inside the silhouette saturates after a few entries."""
    return _edit(synth_rows(n, seed, 1.0, 0.01, 0.08), on_sphere=radius, alpha_min=240)
