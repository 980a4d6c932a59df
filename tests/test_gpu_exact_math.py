"""The device computes sqrt, 1/x and 1/sqrt(x) with shorter instruction sequences than the compiler's
general correctly rounded expansions (pt_device.h: sqrt_rn, rcp_rn, rcp_sqrt_rn).  Pixel values are
defined through correctly rounded IEEE operations, so these must agree with the general expansions for
EVERY operand: tests/hip/exact_check.hip compares them over all 2^32 binary32 bit patterns on the GPU.
The two control cases prove that the comparison sees a one-ulp difference: the raw v_rcp_f32 / v_sqrt_f32
results differ from the correctly rounded ones for hundreds of millions of patterns."""
import ctypes as C
import os

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def checker():
    so = os.path.join(ROOT, "tests", "hip", "libexact_check.so")
    assert os.path.exists(so), "tests/hip/libexact_check.so is missing: run __graft_entry__.build()"
    L = C.CDLL(so)
    L.exact_check_run.argtypes = [C.c_int, C.POINTER(C.c_ulonglong)]

    def run(which):
        r = (C.c_ulonglong * 2)()
        assert L.exact_check_run(which, r) == 0
        return int(r[0]), int(r[1])
    return run


@pytest.mark.parametrize("which,name", [(0, "sqrt"), (1, "reciprocal"), (2, "reciprocal square root")])
def test_short_sequences_equal_the_general_expansions_for_every_operand(checker, which, name):
    mismatches, first = checker(which)
    assert mismatches == 0, f"{name}: {mismatches} operands differ, the smallest is bit pattern {first:#010x}"


def test_the_comparison_detects_one_ulp(checker):
    assert checker(3)[0] > 10 ** 8 and checker(4)[0] > 10 ** 8
