"""A bounded, seeded slice of tools/gpu_fuzz.py under -m gpu: random pattern sets (2 letters ... all
bytes ... UTF-8 with 2/3/4-byte characters; duplicates, nested patterns) x random haystacks (uniform,
text-like, planted, dense; aligned and unaligned) x all match kinds x overlapping x code points x
both scan kernels, HIP path against the oracle.  Fixed seed, ~60 s; the sequence of cases is
deterministic, the time budget only decides where it stops."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_fuzz_slice_seeded():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import gpu_fuzz
    cases, fails = gpu_fuzz.fuzz(float(os.environ.get("ACX_FUZZ_SECONDS", "60")), 20260927, max_size_log2=22.0,
                                 save_failures=False)
    assert fails == 0, f"{fails} of {cases} cases differ from the oracle"
    assert cases >= 20
