"""The CPU pin -- restatement == the reference itself -- re-run WHERE THE GPU PARITY IS SHOWN.

Most GPU parity tests compare the HIP path with the plain-C restatements (oracle/karto_oracle.c, oracle/hector_oracle.c).
What pins those restatements to the reference's own compiled code is tests/test_oracle_vs_ref.py, a CPU test the round-end
GPU run (`pytest -m gpu`) never executes.  This module re-runs a ten-second subset of it under the `gpu` marker -- the
SURVEY 8(c) known answers, MatchScan bit for bit, the lookup tables / response sums, Hector's updateByScan and its
Gauss-Newton matcher against the reference's unmodified headers -- against the prebuilt oracle/_ref libraries that travel
with the snapshot, so the same GPUTEST record shows both halves of the chain: HIP == restatement == reference."""
import pytest

import test_oracle_vs_ref as pin

pytestmark = pytest.mark.gpu

po = pin.po      # module-scoped fixtures of the CPU pin, re-exported
hpo = pin.hpo


def test_pin_survey_known_answers(po):
    pin.test_survey_known_answers(po)


def test_pin_match_scan_bit_exact(po, workload):
    pin.test_match_scan_bit_exact(po, workload)


def test_pin_tables_probs_and_response_sums(po, workload):
    pin.test_tables_probs_and_response_sums(po, workload)


def test_pin_hector_update_by_scan_vs_reference(hpo):
    pin.test_hector_update_by_scan_vs_reference(hpo, 0.9, 1000, 0.05)


def test_pin_hector_hessian_and_level_match_vs_reference(hpo):
    pin.test_hector_hessian_and_level_match_vs_reference(hpo)


def test_pin_hector_pyramid_match_and_update_vs_reference(hpo):
    pin.test_hector_pyramid_match_and_update_vs_reference(hpo)
