"""ORT-RNG-v2 on the host (numpy), for the few draws the host itself consumes
(the image-source histogram rounding, reference src/sourceMod.f90:396-407).

    base = mix64(seed ^ (GOLDEN * phase));  c = (ray << 24) + k
    h    = mix64(base + GOLDEN * ((c >> 1) + 1))         (mix64 = SplitMix64 finaliser)
    u    = (k even ? h >> 32 : h & 0xffffffff) * 2^-32   (one hash serves two consecutive draws)

ORT-RNG-v2w (kernel variant bit 5, `wide=True`): one hash per draw and 53 bits of it,
    h = mix64(base + GOLDEN * (c + 1)),  u = (h >> 11) * 2^-53

Same definitions as csrc/ort_device.h (device) — restated, not shared.
"""
import numpy as np

GOLDEN = np.uint64(0x9E3779B97F4A7C15)


def mix64(z):
    z = np.asarray(z, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = z ^ (z >> np.uint64(30))
        z = z * np.uint64(0xBF58476D1CE4E5B9)
        z = z ^ (z >> np.uint64(27))
        z = z * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z


def uniforms(seed: int, phase: int, ray: int, draws, wide: bool = False) -> np.ndarray:
    """u for draw indices `draws` (array) of key (seed, phase, ray)."""
    k = np.asarray(draws, dtype=np.uint64)
    with np.errstate(over="ignore"):
        base = mix64(np.uint64(seed & 0xFFFFFFFFFFFFFFFF) ^ (GOLDEN * np.uint64(phase)))
        c = (np.uint64(ray) << np.uint64(24)) + k
        if wide:
            return (mix64(base + GOLDEN * (c + np.uint64(1))) >> np.uint64(11)).astype(np.float64) * 2.0 ** -53
        h = mix64(base + GOLDEN * ((c >> np.uint64(1)) + np.uint64(1)))
    w = np.where((c & np.uint64(1)) != 0, h & np.uint64(0xFFFFFFFF), h >> np.uint64(32))
    return w.astype(np.float64) * 2.0 ** -32
