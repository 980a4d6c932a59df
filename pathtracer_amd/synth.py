"""In-process generator for the large synthetic scenes (1 M / 10 M spheres on a plane).

The distribution follows the reference's scene generator (src/bin/gen_rand_scene.rs:9-119): four
fixed spheres, then one small sphere per grid cell with a material picked by quartile.  The reference
draws from an unseeded thread_rng; here every cell consumes a fixed block of nine outputs of one
splitmix64 stream, which makes the scene a pure function of the requested size and lets numpy build it
without a Python loop:

    stream k-th output = mix(seed + (k+1) * 0x9E3779B97F4A7C15),  u = (out >> 40) * 2^-24
    cell (a, b), a, b in [-s, s), row-major index c = (a + s) * 2s + (b + s), uses outputs 9c .. 9c+8:
      u0 material selector, u1 / u2 position jitter, u3..u8 material parameters
    s = ceil(sqrt(n / pi)); a cell is kept iff |pos| < s            (gen_rand_scene.rs:78)
"""
import numpy as np

from .renderer import SPHERE_DT, VEC16_DT, CameraSettings, Scene

_GAMMA = np.uint64(0x9E3779B97F4A7C15)


def _splitmix64(seed, k):
    """k-th (0-based) output of a splitmix64 stream started at `seed`; k is a uint64 array."""
    with np.errstate(over="ignore"):
        z = np.uint64(seed) + (k + np.uint64(1)) * _GAMMA
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def _u01(seed, k):
    return ((_splitmix64(seed, k) >> np.uint64(40)).astype(np.float32) * np.float32(2.0 ** -24)).astype(np.float32)


def plane_scene(n_target, seed=None):
    """Scene with about n_target small spheres (plus the four fixed ones of gen_rand_scene.rs:32-66)."""
    n_target = int(n_target)
    seed = (0x5EED0000 + n_target) if seed is None else int(seed)
    s = int(np.ceil(np.sqrt(n_target / np.pi)))
    side = 2 * s
    cells = np.arange(side * side, dtype=np.uint64)
    a = (cells // np.uint64(side)).astype(np.int64) - s
    b = (cells % np.uint64(side)).astype(np.int64) - s
    base = cells * np.uint64(9)
    u = [_u01(seed, base + np.uint64(j)) for j in range(9)]
    px = (a.astype(np.float32) + np.float32(0.9) * u[1]).astype(np.float32)
    pz = (b.astype(np.float32) + np.float32(0.9) * u[2]).astype(np.float32)
    py = np.float32(0.2)
    keep = np.sqrt(px * px + py * py + pz * pz) < np.float32(s)
    px, pz = px[keep], pz[keep]
    u = [x[keep] for x in u]
    mat = u[0]
    is_lam = mat < 0.25
    is_met = (mat >= 0.25) & (mat < 0.5)
    is_gla = (mat >= 0.5) & (mat < 0.75)
    n_small = len(px)

    # fixed spheres: ground (radius grows with the field), glass, lambertian, metal
    ground_r = np.float32(1000.0 * max(1.0, s / 10.0))
    spheres = np.zeros(4 + n_small, SPHERE_DT)
    spheres[0] = ((0.0, -ground_r, 0.0), ground_r, 1, 0, (0, 0))
    spheres[1] = ((0.0, 1.0, 0.0), 1.0, 3, 0, (0, 0))
    spheres[2] = ((-4.0, 1.0, 0.0), 1.0, 1, 1, (0, 0))
    spheres[3] = ((4.0, 1.0, 0.0), 1.0, 2, 0, (0, 0))

    lam = np.zeros(2 + int(is_lam.sum()), VEC16_DT)
    lam["v"][0] = (0.5, 0.5, 0.5)
    lam["v"][1] = (0.4, 0.2, 0.1)
    lam["v"][2:, 0] = u[3][is_lam] * u[4][is_lam]
    lam["v"][2:, 1] = u[5][is_lam] * u[6][is_lam]
    lam["v"][2:, 2] = u[7][is_lam] * u[8][is_lam]
    met = np.zeros(1 + int(is_met.sum()), VEC16_DT)
    met["v"][0] = (0.7, 0.6, 0.5)
    for c in range(3):
        met["v"][1:, c] = np.float32(0.5) + np.float32(0.5) * u[3 + c][is_met]
    met["w"][1:] = np.float32(0.5) * u[6][is_met]
    gla = np.zeros(1 + int(is_gla.sum()), np.float32)
    gla[0] = 1.5
    gla[1:] = np.float32(1.0) + np.float32(3.0) * u[3][is_gla]
    lights = np.zeros(1, VEC16_DT)
    lights["v"][0] = (1.0, 1.0, 1.0)

    small = spheres[4:]
    small["pos"][:, 0] = px
    small["pos"][:, 1] = py
    small["pos"][:, 2] = pz
    small["rad"] = 0.2
    mat_type = np.zeros(n_small, np.uint32)          # light by default (mat >= 0.75)
    mat_index = np.zeros(n_small, np.uint32)
    mat_type[is_lam] = 1
    mat_index[is_lam] = 2 + np.arange(int(is_lam.sum()), dtype=np.uint32)
    mat_type[is_met] = 2
    mat_index[is_met] = 1 + np.arange(int(is_met.sum()), dtype=np.uint32)
    mat_type[is_gla] = 3
    mat_index[is_gla] = 1 + np.arange(int(is_gla.sum()), dtype=np.uint32)
    small["mat_type"] = mat_type
    small["mat_index"] = mat_index

    # camera of gen_rand_scene.rs:12-23: from (13, 2, 3) towards the origin, vfov 20
    look_from = np.array([13.0, 2.0, 3.0], np.float32)
    d = -look_from / np.float32(np.sqrt(np.float32((look_from * look_from).sum())))
    pitch = np.degrees(np.arcsin(d[1]))
    yaw = np.degrees(np.arctan2(d[0], d[2]))
    cam = CameraSettings(tuple(look_from), np.float32(yaw), np.float32(pitch), np.float32(20.0))
    return Scene(cam, spheres, lights, lam, met, gla)
