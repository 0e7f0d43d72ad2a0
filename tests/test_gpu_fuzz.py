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


def test_corner_cases_match_oracle(oracle_port):
    """tiny sets, sample budgets of 0 ... 600, thresholds 0 ... 100 px, confidences 0 ... 1, noise-free / pixel-quantised coordinates,
    repeated correspondences (tools/gpu_fuzz.py edges)"""
    import gpu_fuzz
    assert gpu_fuzz.run_edges(400, 77) == (0, 0)


def test_random_batches_match_oracle_pair_by_pair(oracle_port):
    """ragged batches through the batch entry points with random variants, grid caps, set-aside thresholds and per-call scheduling flags"""
    import gpu_fuzz
    assert gpu_fuzz.run_batches(12, 3) == (0, 0)


def test_legacy_drivers_random_cases(oracle_port):
    import gpu_fuzz
    assert gpu_fuzz.run_legacy(150, 5) == (0, 0)
