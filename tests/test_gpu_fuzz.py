"""A fixed slice of tools/gpu_fuzz.py in the GPU suite: 80 seeded random batches (shape, content, CDC parameters) through the
C ABI against the oracle, every column.  The open-ended form ran 1 680 cases / 23.6 M chunks in two runs without a difference
(profiles/r04_gpu_fuzz.txt)."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_seeded_random_batches_match_the_oracle(oracle):
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import gpu_fuzz
    chunks = 0
    for case in range(80):
        desc, bad, n = gpu_fuzz.one_case(np.random.default_rng([2024, case]), case)
        assert not bad, "seed 2024 case %d (%s): %s" % (case, desc, ", ".join(bad))
        chunks += n
    assert chunks > 100000
