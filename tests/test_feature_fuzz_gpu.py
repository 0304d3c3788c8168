"""tools/feature_fuzz.py as a test: 140 random cases (shapes, transform sizes, hops, windows, pad modes, ragged batches, speeds) through
stft / fbank / the Kaldi fbank / istft / mfcc / compute_deltas / the resampler against the oracle (scipy for the resampler), at the
tolerances of the pinned parity tests.  The pinned tests hold chosen points; this walks the space between them (round 6: 2 100 cases
over three seeds, worst case at 0.20 of its tolerance - stft; the last hop samples of an istft divide by the vanishing tail of the
window envelope and are held at 5e-3 instead of 2e-5: the float64 oracle's own round trip is off by 4e-5 there)."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu


def test_random_feature_cases_stay_inside_the_pinned_tolerances():
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from tools.feature_fuzz import run

    res = run(cases=140, seed=20261004)
    assert res["n_failures"] == 0, res
    assert set(res["cases"]) == {"stft", "fbank", "kaldi", "istft", "mfcc", "deltas", "resample"}
