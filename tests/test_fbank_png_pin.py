"""A coarse pin of the oracle's features.fbank against something MindSpore itself produced: the figure of tutorial cell 11
(`features.fbank(wav, n_fft=512)` on the reference's sample wav), decoded to clip(dB, 0) / max on its pixel grid by
tests/golden/gen_fbank_png_pin.py (build container; only the decoded array is committed).  SURVEY 8(c): MindSpore's Spectrogram /
MelScale cannot run here, so mel / fbank parity is otherwise 'vs our restatement' only.  What this catches: a wrong power (magnitude
vs power), log base or reference level only through the >0 dB area (the figure is normalised by its maximum), a wrong window /
n_fft normalisation (tens of dB), a wrong hop (375 frames), a wrong mel scale or filter shape (row profile)."""
import os

import numpy as np
import pytest
from scipy.ndimage import map_coordinates

HERE = os.path.dirname(os.path.abspath(__file__))


def _on_pixel_grid(mat, shape):
    """clip(mat, 0) / max resampled (bilinear ~ gouraud) at the pixel centres of an axes area whose x spans the frames and whose y
    spans the mel bins, low mel at the bottom."""
    n = np.clip(mat, 0, None)
    n = n / n.max() if n.max() > 0 else n
    h, w = shape
    rr = (h - 1 - np.arange(h)) / (h - 1) * (mat.shape[0] - 1)
    cc = np.arange(w) / (w - 1) * (mat.shape[1] - 1)
    r, c = np.meshgrid(rr, cc, indexing="ij")
    return map_coordinates(n, [r, c], order=1)


@pytest.fixture(scope="module")
def pin():
    z = np.load(os.path.join(HERE, "golden", "fbank_tutorial_png.npz"))
    return z["value_index"].astype(np.float64) / 255.0


@pytest.fixture(scope="module")
def fbank_db():
    from mindaudio_amd.data import io as mio
    from oracle import speech_features as O

    wav, sr = mio.read(os.path.join(HERE, "golden", "BAC009S0002W0122.wav"))
    assert sr == 16000
    m = O.fbank(wav, n_fft=512)
    assert m.shape == (40, 375)  # the tutorial's printed shape
    return m


def _scores(p, q):
    return (np.corrcoef(p.ravel(), q.ravel())[0, 1], np.corrcoef(p.sum(1), q.sum(1))[0, 1], np.corrcoef(p.sum(0), q.sum(0))[0, 1])


def test_oracle_fbank_matches_the_tutorial_figure(pin, fbank_db):
    q = _on_pixel_grid(fbank_db, pin.shape)
    corr, rows, cols = _scores(pin, q)
    assert corr >= 0.90 and rows >= 0.97 and cols >= 0.92, (corr, rows, cols)
    # the area above 0 dB (vmin = 0 is absolute): 1.2 % of the figure, 1.5 % of the oracle's matrix on the same grid
    fp, fq = float((pin > 0.1).mean()), float((q > 0.1).mean())
    assert 0.6 * fp <= fq <= 1.6 * fp, (fp, fq)
    # the strongest blob sits at the same mel row and within three frames
    (rp, cp), (rq, cq) = np.unravel_index(pin.argmax(), pin.shape), np.unravel_index(q.argmax(), q.shape)
    assert abs(int(rp) - int(rq)) <= 3 and abs(int(cp) - int(cq)) <= 3


@pytest.mark.parametrize("name", ["+10 dB", "-10 dB", "hop 160 instead of win/2", "mel bins reversed", "linear (no mel warp)"])
def test_the_pin_rejects_wrong_variants(pin, fbank_db, name):
    """Sensitivity: variants that a wrong reading of the MindSpore semantics would produce do NOT pass the thresholds above."""
    from mindaudio_amd.data import io as mio
    from oracle import speech_features as O

    m = fbank_db
    if name == "+10 dB":
        v = m + 10.0
    elif name == "-10 dB":
        v = m - 10.0
    elif name == "hop 160 instead of win/2":
        wav, _ = mio.read(os.path.join(HERE, "golden", "BAC009S0002W0122.wav"))
        v = O.fbank(wav, n_fft=512, hop_length=160)[:, :375]  # 600 frames: the first 375 stretched over the figure
    elif name == "mel bins reversed":
        v = m[::-1]
    else:  # rows resampled as if the filters were linearly spaced in Hz: undo the mel warp of the row axis
        hz = 700.0 * (10 ** (np.linspace(0, 2595 * np.log10(1 + 8000 / 700.0), 42)[1:-1] / 2595.0) - 1)
        lin = np.linspace(hz[0], hz[-1], 40)
        v = np.stack([np.interp(lin, hz, m[:, t]) for t in range(m.shape[1])], axis=1)
    q = _on_pixel_grid(v, pin.shape)
    corr, rows, cols = _scores(pin, q)
    fp, fq = float((pin > 0.1).mean()), float((q > 0.1).mean())
    ok = corr >= 0.90 and rows >= 0.97 and cols >= 0.92 and 0.6 * fp <= fq <= 1.6 * fp
    assert not ok, (name, corr, rows, cols, fp, fq)
