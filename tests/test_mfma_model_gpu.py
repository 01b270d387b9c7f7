"""The ONE modelled term of the coarse filters' error bound, measured: what v_mfma_f32_16x16x32_bf16's internal
accumulation of 32 bf16 products + C deviates from the exact sum.  Rounds 2-4 priced it at 4 ulp-units of the magnitudes per 16
dimensions (x 1.25) = 10 u (|C| + sum |a_k b_k|) per 32-product instruction, unmeasured; this test and a wider probe found up to 8.8 -- terms far below the
largest one are TRUNCATED, in two stages (3.8 u when the large value is C, 8.8 u when it is a product) -- and csrc/flat_collect.h
CL_MFMA_UNITS now charges 8 per 16 dimensions: 16 u per instruction, 20 with the safety factor.  The instruction is run bare (C ABI
mvs_debug_mfma_bf16_16x16x32, csrc/util_kernels.hip mfma_bf16_probe_kernel) on tiles built to hurt: wide exponent ranges, heavy
cancellation, C far above / far below the products, products at the bottom of the accumulator's alignment window.  Products of two
bf16 numbers have 16-bit significands and the exact sum of 33 such terms fits a float64 here, so numpy's float64 IS the oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
U = 2.0**-24


@pytest.fixture(scope="module")
def mf():
    import mi355_faiss

    return mi355_faiss


def to_bf16_bits(x):
    """round-to-nearest-even to bfloat16, as uint16 bit patterns"""
    b = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    return (((b + 0x7FFF + ((b >> 16) & 1)) >> 16) & 0xFFFF).astype(np.uint16)


def from_bf16_bits(b):
    return (b.astype(np.uint32) << 16).view(np.float32)


def tiles(kind, n, rs):
    A = rs.randn(n, 16, 32).astype(np.float32)
    B = rs.randn(n, 16, 32).astype(np.float32)  # Bt: the columns of B as rows
    C = rs.randn(n, 16, 16).astype(np.float32)
    if kind == "wide_exponents":
        A *= np.exp2(rs.randint(-12, 13, size=A.shape)).astype(np.float32)
        B *= np.exp2(rs.randint(-12, 13, size=B.shape)).astype(np.float32)
    elif kind == "cancellation":  # pairs of products that cancel almost exactly, a small C survives
        A[:, :, 1::2] = -A[:, :, 0::2]
        B[:, :, 1::2] = B[:, :, 0::2] * (1.0 + rs.choice([0.0, 2.0**-7, -(2.0**-7)], size=B[:, :, 0::2].shape)).astype(np.float32)
        C *= 1e-3
    elif kind == "big_c":  # the chain start (beta + gamma in the kernels) dwarfs the products
        C *= 2.0**14
    elif kind == "small_c":
        C *= 2.0**-20
    elif kind == "one_big_term":  # one product 2^20 above the rest: the others sit at the bottom of the alignment window
        A[:, :, 0] *= 2.0**10
        B[:, :, 0] *= 2.0**10
    elif kind == "same_sign":  # no cancellation at all: the sum is as large as the magnitudes
        A, B, C = np.abs(A), np.abs(B), np.abs(C)
    elif kind in ("big_product_small_terms", "big_c_small_terms"):  # everything else sits just below the alignment cut of the large value
        A = rs.uniform(0.0, 0.25, size=(n, 16, 32)).astype(np.float32) * rs.choice([-1.0, 1.0], size=(n, 16, 32)).astype(np.float32)
        B = np.ones((n, 16, 32), dtype=np.float32)
        C = np.zeros((n, 16, 16), dtype=np.float32)
        if kind == "big_c_small_terms":
            C[:] = 2.0**20
        else:
            A[:, :, 0] = 2.0**10
            B[:, :, 0] = 2.0**10
    elif kind == "log_uniform":  # magnitudes spread log-uniformly over 16 binades, random signs
        A = (rs.choice([-1.0, 1.0], size=(n, 16, 32)) * np.exp2(rs.uniform(-8, 8, size=(n, 16, 32)))).astype(np.float32)
        B = (rs.choice([-1.0, 1.0], size=(n, 16, 32)) * np.exp2(rs.uniform(-4, 4, size=(n, 16, 32)))).astype(np.float32)
        C = (rs.choice([-1.0, 1.0], size=(n, 16, 16)) * np.exp2(rs.uniform(-8, 8, size=(n, 16, 16)))).astype(np.float32)
    return A, B, C


@pytest.mark.parametrize("kind", ["plain", "wide_exponents", "cancellation", "big_c", "small_c", "one_big_term", "same_sign",
                                  "big_product_small_terms", "big_c_small_terms", "log_uniform"])
def test_bf16_mfma_accumulation_error_is_inside_the_modelled_units(mf, kind):
    rs = np.random.RandomState(sum(map(ord, kind)))
    n = 4096
    A, B, C = tiles(kind, n, rs)
    Ab, Bb = to_bf16_bits(A), to_bf16_bits(B)
    D = mf.mfma_bf16_16x16x32(Ab, Bb, C)
    a = from_bf16_bits(Ab).astype(np.float64)
    b = from_bf16_bits(Bb).astype(np.float64)
    prod = np.einsum("tik,tjk->tijk", a, b)  # [tile][row][column][k], exact in float64
    exact = prod.sum(-1) + C.astype(np.float64)
    mag = np.abs(prod).sum(-1) + np.abs(C.astype(np.float64))
    err = np.abs(D.astype(np.float64) - exact)
    ratio = err / (U * np.maximum(mag, 1e-300))
    worst = float(ratio.max())
    # the model: CL_MFMA_UNITS = 8 ulp-units of the magnitudes per 16 dimensions = 16 per 32-product instruction (the bound takes x 1.25)
    assert worst <= 16.0, (kind, worst)
    print(f"\n[mfma model] {kind}: worst |D - exact| = {worst:.3f} u (|C| + sum |a b|) over {n * 256} results (charged: 16, with the safety factor 20)")
