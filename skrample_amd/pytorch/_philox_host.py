"""Host-side Philox4x32-10 (numpy), used only for the few scalars a generator needs *before* launching
(the per-sample pyramid level geometry).  Same algorithm and counter layout as csrc/skr_philox.h:
key = 64-bit seed, counter = (block lo, block hi, stream lo, stream hi)."""

from __future__ import annotations

import numpy as np

_M0, _M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
_W0, _W1 = np.uint32(0x9E3779B9), np.uint32(0xBB67AE85)
_LO = np.uint64(0xFFFFFFFF)


def philox_u32(seeds: np.ndarray, stream: int, n_blocks: int) -> np.ndarray:
    "uint32 [len(seeds), n_blocks*4]: blocks 0..n_blocks-1 of `stream` for every seed"
    seeds = np.asarray(seeds, dtype=np.uint64).reshape(-1, 1)
    blocks = np.arange(n_blocks, dtype=np.uint64).reshape(1, -1)
    c = [
        np.broadcast_to((blocks & _LO).astype(np.uint32), (seeds.shape[0], n_blocks)).copy(),
        np.broadcast_to((blocks >> np.uint64(32)).astype(np.uint32), (seeds.shape[0], n_blocks)).copy(),
        np.full((seeds.shape[0], n_blocks), stream & 0xFFFFFFFF, dtype=np.uint32),
        np.full((seeds.shape[0], n_blocks), (stream >> 32) & 0xFFFFFFFF, dtype=np.uint32),
    ]
    k0 = np.broadcast_to((seeds & _LO).astype(np.uint32), c[0].shape).copy()
    k1 = np.broadcast_to((seeds >> np.uint64(32)).astype(np.uint32), c[0].shape).copy()
    with np.errstate(over="ignore"):
        for _ in range(10):
            p0 = c[0].astype(np.uint64) * _M0
            p1 = c[2].astype(np.uint64) * _M1
            c = [
                (p1 >> np.uint64(32)).astype(np.uint32) ^ c[1] ^ k0,
                (p1 & _LO).astype(np.uint32),
                (p0 >> np.uint64(32)).astype(np.uint32) ^ c[3] ^ k1,
                (p0 & _LO).astype(np.uint32),
            ]
            k0 = (k0 + _W0).astype(np.uint32)
            k1 = (k1 + _W1).astype(np.uint32)
    return np.stack(c, axis=-1).reshape(seeds.shape[0], n_blocks * 4)


def uniform01(seeds: np.ndarray, stream: int, count: int) -> np.ndarray:
    "float32-exact uniforms in [0, 1): (x >> 8) * 2^-24, shape [len(seeds), count]"
    words = philox_u32(seeds, stream, (count + 3) // 4)[:, :count]
    return (words >> np.uint32(8)).astype(np.float64) * 2.0**-24
