"""Integer identities the kernels rely on, checked exhaustively on the host (no GPU)."""
import numpy as np


def test_multiply_high_division_is_exact_for_the_prepass_item_range():
    """sdx_kernels.h small_div: k / d == (k * ceil(2^32 / d)) >> 32 for 0 <= k < 65536, 1 < d < 65536 (the pre-pass indexes at most
    kPreLines * kPreDepths = 2048 items by divisors <= 64; d = 1 is the identity there)."""
    k = np.arange(65536, dtype=np.uint64)
    for d in list(range(2, 2050)) + [4097, 32768, 65535]:
        magic = np.uint64(0xFFFFFFFF // d + 1)
        assert np.array_equal((k * magic) >> np.uint64(32), k // np.uint64(d)), d
