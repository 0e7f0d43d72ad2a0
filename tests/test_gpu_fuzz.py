"""GPU: randomised sweep over F / H, every metric, LAF on/off, degenerate scenes, n = 8..3000, both kernel variants and
all placement modes: results AND the sample / LO counters equal the oracle's (tools/gpu_fuzz.py runs longer sweeps)."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))


def test_random_cases_match_oracle_results_and_counters(oracle_port):
    import gpu_fuzz
    bad_results, bad_counters = gpu_fuzz.run(120, 2024, verbose=True)
    assert bad_results == 0 and bad_counters == 0
