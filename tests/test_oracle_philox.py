"""Philox4x32-10 known-answer vectors (Random123 kat_vectors) for the oracle's generator."""
import ctypes


KAT = [
    ((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
    ((0xffffffff,) * 4, (0xffffffff, 0xffffffff), (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
    ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
     (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)),
]


def test_philox_kat(oracle_lib):
    for ctr, key, exp in KAT:
        c = (ctypes.c_uint32 * 4)(*ctr)
        k = (ctypes.c_uint32 * 2)(*key)
        o = (ctypes.c_uint32 * 4)()
        oracle_lib.orc_philox4x32_10(c, k, o)
        assert tuple(o) == exp
