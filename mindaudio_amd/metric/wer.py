"""Word / character error rate of one utterance — same contract as mindaudio.metric.wer.wer (metric/wer.py:4-57):
(deletions + insertions + substitutions) / len(ref) by edit distance; ValueError on an empty reference.
Pure host logic (scoring of decoded token lists)."""


def wer(ref, hyp):
    if not ref:
        raise ValueError("The reference utterance must not be empty.")
    # rolling one-row Levenshtein table: prev[j] = distance(ref[:i-1], hyp[:j])
    prev = list(range(len(hyp) + 1))
    for i, r in enumerate(ref, 1):
        cur = [i] + [0] * len(hyp)
        for j, h in enumerate(hyp, 1):
            cur[j] = prev[j - 1] if r == h else 1 + min(prev[j - 1], cur[j - 1], prev[j])
        prev = cur
    return prev[-1] / len(ref)
