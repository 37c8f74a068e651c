"""Philox4x32-10 (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3", SC'11) in numpy.

TEST INFRASTRUCTURE (checker of the device generator in phoregen_amd/csrc/posterior.hip; never imported by the product).
The reference's sampler draws from torch's global generator (models/common.py:425-431, models/transition.py:60); a
counter-based generator replaces it on the device, so what is pinned here is the generator itself: the three
known-answer vectors of Random123's kat_vectors file (philox4x32 10) are checked in tests/test_host_cpu.py.
"""
import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = np.uint32(0x9E3779B9), np.uint32(0xBB67AE85)

# (counter c0..c3, key k0 k1) -> output, Random123 kat_vectors "philox4x32 10"
KAT = [
    ((0x00000000,) * 4, (0x00000000,) * 2, (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
    ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
    ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0), (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)),
]


def philox4x32(ctr, key, rounds=10):
    """ctr [n,4] uint32, key [n,2] uint32 -> [n,4] uint32."""
    c = np.asarray(ctr, dtype=np.uint32).reshape(-1, 4).copy()
    k = np.asarray(key, dtype=np.uint32).reshape(-1, 2).copy()
    with np.errstate(over='ignore'):
        for _ in range(rounds):
            p0 = M0 * c[:, 0].astype(np.uint64)
            p1 = M1 * c[:, 2].astype(np.uint64)
            n0 = (p1 >> np.uint64(32)).astype(np.uint32) ^ c[:, 1] ^ k[:, 0]
            n2 = (p0 >> np.uint64(32)).astype(np.uint32) ^ c[:, 3] ^ k[:, 1]
            c = np.stack([n0, p1.astype(np.uint32), n2, p0.astype(np.uint32)], 1)
            k = np.stack([k[:, 0] + W0, k[:, 1] + W1], 1)
    return c


def uniform24(words):
    """The device's word -> [0,1) map (posterior.hip Philox::uniform): top 24 bits / 2^24."""
    return (np.asarray(words, dtype=np.uint32) >> np.uint32(8)).astype(np.float32) * np.float32(1.0 / 16777216.0)
