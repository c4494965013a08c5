"""Philox4x32 known-answer vectors (Random123 kat_vectors: ten rounds, and the seven rounds of the per-base draws) for the
oracle's generator."""
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


KAT7 = [
    ((0, 0, 0, 0), (0, 0), (0x5f6fb709, 0x0d893f64, 0x4f121f81, 0x4f730a48)),
    ((0xffffffff,) * 4, (0xffffffff, 0xffffffff), (0x5207ddc2, 0x45165e59, 0x4d8ee751, 0x8c52f662)),
    ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
     (0x4dfccaba, 0x190a87f0, 0xc47362ba, 0xb6b5242a)),
]


def test_philox_seven_rounds_kat(oracle_lib):
    """The per-base draws (KIND_BASE) take kBaseRounds = 7 rounds (oracle/philox.h, sg_device.h)."""
    oracle_lib.orc_philox4x32_r.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_uint32),
                                            ctypes.POINTER(ctypes.c_uint32)]
    assert oracle_lib.orc_base_rounds() == 7
    for ctr, key, exp in KAT7:
        c = (ctypes.c_uint32 * 4)(*ctr)
        k = (ctypes.c_uint32 * 2)(*key)
        o = (ctypes.c_uint32 * 4)()
        oracle_lib.orc_philox4x32_r(7, c, k, o)
        assert tuple(o) == exp
