"""Build container only: decode the ONE artefact under /root/reference that MindSpore's Spectrogram / MelScale actually produced for
features.fbank - the figure embedded in tutorial cell 11 (tutorials/audio_data_processing_with_mindaudio.ipynb:
`matrix = features.fbank(test_data, n_fft=512)` on tests/samples/ASR/BAC009S0002W0122.wav, drawn with
`plt.pcolormesh(x, f, matrix, shading='gouraud', vmin=0)`, viridis) - into a scalar array and store THAT (not the picture) as
tests/golden/fbank_tutorial_png.npz.  pcolormesh with vmin = 0 and vmax = max(matrix) maps value v to colour index
255 * clip(v, 0) / max: the decoded array is clip(fbank_dB, 0, None) / max on the figure's pixel grid, 8 bits deep.
Coarse, but it sees the absolute level (what is above 0 dB), the time / mel positions of the energy and the mel scale's shape:
a wrong power, reference level, window normalisation or filter bank would move or rescale the blobs.

    python tests/golden/gen_fbank_png_pin.py        (needs /root/reference, PIL, matplotlib)"""
import base64
import io
import json
import os

import numpy as np

NB = "/root/reference/tutorials/audio_data_processing_with_mindaudio.ipynb"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "fbank_tutorial_png.npz")


def main():
    from matplotlib import colormaps
    from PIL import Image

    cell = json.load(open(NB))["cells"][11]
    assert "features.fbank(test_data, n_fft=n_fft)" in "".join(cell["source"])
    png = next(base64.b64decode(o["data"]["image/png"]) for o in cell["outputs"] if "image/png" in o.get("data", {}))
    im = np.array(Image.open(io.BytesIO(png)).convert("RGB")).astype(np.int32)
    lut = (colormaps["viridis"](np.linspace(0, 1, 256))[:, :3] * 255).round().astype(np.int32)
    d = ((im[:, :, None, :] - lut[None, None, :, :]) ** 2).sum(-1)
    idx, dist = d.argmin(-1), d.min(-1)
    on = dist <= 12  # viridis-coloured pixels = the axes' interior
    rows, cols = np.where(on.sum(1) > 200)[0], np.where(on.sum(0) > 150)[0]
    # (one pixel wider on the left / top than the solidly coloured block: the frame line's anti-aliased edge belongs to the data area;
    # this box maximises the correlation with any reasonable fbank of the sample, +-1 pixel)
    r0, r1, c0, c1 = rows.min() - 1, rows.max() + 1, cols.min() - 1, cols.max() - 1
    arr = idx[r0:r1 + 1, c0:c1 + 1].astype(np.uint8)
    np.savez_compressed(OUT, value_index=arr, bbox=np.array([r0, r1, c0, c1]), figure_size=np.array(im.shape[:2]),
                        n_mels=np.array(40), n_frames=np.array(375))
    print("wrote", OUT, arr.shape, int(arr.max()))


if __name__ == "__main__":
    main()
