"""Mirror of examples/conformer/asr_model.py — forward / evaluation loss on MI355X, pure CTC (ctc_weight = 1.0, decoder None:
asr_model.py:327-328) and the hybrid CTC / attention configuration of the shipped conformer.yaml.

ASRModel.forward takes the 11 columns of the reference batch in the reference order (train.py:38-50) and returns
(loss, acc_att) like ASRModelWithAcc.construct (asr_model.py:75-153)."""
import torch
import torch.nn as nn

from .. import ops
from ..models.conformer import ConformerEncoder


class CTC(nn.Module):
    """mindaudio/loss/ctc_loss.py:10-64: Dense(eprojs -> odim), float32 log_softmax, CTCLossV2(blank 0,
    zero_infinity), sum / batch."""

    def __init__(self, odim, encoder_output_size, dropout_rate=0.0, compute_type=None):
        super().__init__()
        self.ctc_lo = nn.Linear(encoder_output_size, odim)
        self._w = None
        self.register_load_state_dict_post_hook(lambda module, _keys: setattr(module, "_w", None))  # stale bf16 copy

    @torch.no_grad()
    def prepare(self):
        self._w = (self.ctc_lo.weight.detach().to(torch.bfloat16).contiguous(),
                   self.ctc_lo.bias.detach().float().contiguous())
        return self

    @torch.no_grad()
    def logits(self, hs_pad):
        if self._w is None:
            self.prepare()
        b, t, d = hs_pad.shape
        a = ops.cast_bf16(hs_pad.reshape(b * t, d).contiguous())
        return ops.gemm(a, self._w[0], bias=self._w[1], out_dtype=torch.float32)

    @torch.no_grad()
    def forward(self, hs_pad, hlens, ys_pad, ys_lengths):
        b, t, _ = hs_pad.shape
        loss, _ = ops.ctc_loss(self.logits(hs_pad), b, t, ys_pad, hlens, ys_lengths, blank=0, zero_infinity=True)
        return loss


class ASRModel(nn.Module):
    """Encoder + CTC [+ attention decoder] (asr_model.py:16-153).  forward() is the evaluation forward (create_asr_eval_net,
    asr_model.py:355-371) of either configuration: pure CTC (ctc_weight == 1.0) or the hybrid 0.3 / 0.7 loss with the
    TransformerDecoder and label smoothing; the training step (forward + backward + optimizer) is mindaudio_amd.train.engine."""

    def __init__(self, vocab_size, encoder, ctc, ctc_weight=1.0, decoder=None, lsm_weight=0.0, reverse_weight=0.0,
                 length_normalized_loss=False):
        super().__init__()
        if ctc_weight != 1.0 and decoder is None:
            raise ValueError("ctc_weight != 1.0 needs a decoder (asr_model.py:327-337)")
        if reverse_weight != 0.0:
            # asr_model.py:175-178 weighs in a right-to-left decoder's loss, but the reference ships no such decoder:
            # TransformerDecoder.construct returns a constant as r_decoder_out (models/conformer.py:606-639), so the
            # reference itself fails on reverse_weight > 0.  Same here, with a message.
            raise ValueError("reverse_weight > 0 needs a bidirectional decoder; the reference has none "
                             "(models/conformer.py:620, 639)")
        self.vocab_size, self.encoder, self.ctc, self.ctc_weight = vocab_size, encoder, ctc, ctc_weight
        self.decoder, self.lsm_weight = decoder, lsm_weight
        self.reverse_weight, self.length_normalized_loss = 0.0, bool(length_normalized_loss)

    @torch.no_grad()
    def forward(self, xs_pad, ys_pad, ys_in_pad=None, ys_out_pad=None, r_ys_in_pad=None, r_ys_out_pad=None,
                xs_masks=None, ys_sub_masks=None, ys_masks=None, ys_lengths=None, xs_chunk_masks=None):
        encoder_out, encoder_mask = self.encoder(xs_pad, xs_masks, xs_chunk_masks)
        # asr_model.py:109-114: lengths = mask.squeeze().sum(1) as int32
        encoder_out_lens = encoder_mask.to(torch.float32).reshape(encoder_mask.shape[0], -1).sum(1).to(torch.int32)
        loss_att = acc_att = None
        if self.ctc_weight != 1.0:  # attention-decoder branch (asr_model.py:117-129, 154-209)
            loss_att, acc_att = self._calc_att_loss(encoder_out, encoder_mask, ys_in_pad, ys_out_pad, ys_masks, ys_sub_masks)
        loss_ctc = self.ctc(encoder_out, encoder_out_lens, ys_pad, ys_lengths) if self.ctc_weight != 0.0 else None
        if loss_ctc is None:
            return loss_att, acc_att
        if loss_att is None:
            return loss_ctc, None
        return self.ctc_weight * loss_ctc + (1.0 - self.ctc_weight) * loss_att, acc_att  # asr_model.py:138-139

    @torch.no_grad()
    def _calc_att_loss(self, encoder_out, encoder_mask, ys_in_pad, ys_out_pad, ys_masks, ys_sub_masks):
        """asr_model.py:154-209: decoder scores -> LabelSmoothingLoss (KL, summed, / batch: label_smoothing_loss.py:24-117) and the
        token accuracy over the unmasked positions."""
        from ..train import kernels as K

        if ys_in_pad is None or ys_out_pad is None or ys_masks is None or ys_sub_masks is None:
            raise ValueError("the hybrid loss needs ys_in_pad, ys_out_pad, ys_sub_masks and ys_masks")
        scores, _ = self.decoder(encoder_out, encoder_mask, ys_in_pad, ys_sub_masks)
        b, l1, v = scores.shape
        logits = scores.reshape(b * l1, v)  # a view of the decoder's 64-padded row buffer
        tgt = ys_out_pad.to(torch.int32).contiguous().reshape(-1)
        tmask = ys_masks.to(torch.float32).contiguous().reshape(-1)
        stats, _ = K.label_smoothing_loss_grad(logits, v, tgt, tmask, self.lsm_weight, 0.0)
        # label_smoothing_loss.py:105-106: divided by the token count when normalize_length, else by the batch size
        return stats[0] / (stats[2] if self.length_normalized_loss else b), stats[1] / stats[2]


def create_asr_model(input_dim, vocab_size, encoder_conf=None, global_cmvn=None, ctc_weight=1.0, decoder_conf=None,
                     lsm_weight=0.0, length_normalized_loss=False):
    """creadte_asr_model (asr_model.py:301-352): decoder None when ctc_weight == 1.0, else a TransformerDecoder."""
    from ..models.decoder import TransformerDecoder

    encoder = ConformerEncoder(input_dim, global_cmvn=global_cmvn, **(encoder_conf or {}))
    ctc = CTC(vocab_size, encoder.output_size())
    decoder = None
    if ctc_weight != 1.0:
        decoder = TransformerDecoder(vocab_size, encoder.output_size(), **(decoder_conf or {}))
    return ASRModel(vocab_size, encoder, ctc, ctc_weight, decoder=decoder, lsm_weight=lsm_weight,
                    length_normalized_loss=length_normalized_loss)


class ASREvalNet(nn.Module):
    """create_asr_eval_net (asr_model.py:355-371): all-reduce (SUM) of the loss over the data-parallel ranks, divided
    by device_num.  One process per GPU; the collective is torch.distributed's (backend "nccl" = RCCL over xGMI on the
    GPU box, "gloo" in the CPU tests)."""

    def __init__(self, network, device_num):
        super().__init__()
        self.network, self.device_num = network, device_num

    @torch.no_grad()
    def forward(self, *inputs, **kwargs):
        out = self.network(*inputs, **kwargs)
        loss = out[0] if isinstance(out, tuple) else out
        loss = loss.clone().reshape(1)
        if self.device_num > 1:
            import torch.distributed as dist

            dist.all_reduce(loss, op=dist.ReduceOp.SUM)
        return loss[0] / self.device_num


def shard_batch(columns, rank, group_size):
    """Every rank builds the same global batch and keeps the strided slice batch[rank::group_size]
    (examples/conformer/dataset.py:552-553)."""
    return [c[rank::group_size] for c in columns]


def remove_duplicates_and_blank(hyp):
    """mindaudio/utils/common.py:116-125 on a host list (the device version lives in ma_ctc_greedy_search_f32)."""
    out, prev = [], None
    for tok in hyp:
        if tok != prev and tok != 0:
            out.append(tok)
        prev = tok
    return out


class CTCGreedySearch(nn.Module):
    """CTC greedy search net (models/decoders/decoder_factory.py:9-56): forward(xs_pad, xs_masks, xs_lengths) ->
    (topk_index (B, T') int32 with padded frames zeroed, topk_prob (B, T') float32 log-probabilities).
    `xs_masks` is the un-subsampled (B, 1, T) pad mask, sliced [:, :, :-2:2][:, :, :-2:2] here as in the reference."""

    def __init__(self, backbone, pretrained_model=False):
        super().__init__()
        if pretrained_model:
            raise NotImplementedError("wav2vec front ends are outside the built path")
        self.backbone = backbone

    @torch.no_grad()
    def forward(self, xs_pad, xs_masks, xs_lengths=None):
        best, logp, _, _ = self._search(xs_pad, xs_masks)
        return best, logp

    @torch.no_grad()
    def _search(self, xs_pad, xs_masks):
        sub = xs_masks[:, :, :-2:2][:, :, :-2:2].contiguous()
        enc, enc_mask = self.backbone.encoder(xs_pad, sub, sub)
        b, t2, _ = enc.shape
        logits = self.backbone.ctc.logits(enc)
        return ops.ctc_greedy_search(logits, b, t2, logits.shape[1], enc_mask.reshape(-1).to(torch.float32).contiguous())


def ctc_greedy_search(model, xs_pad, xs_masks, xs_lengths=None):
    """utils/recognize.py:254-270: (hyps: list of token lists with repeats and blanks removed, scores: per-utterance
    maximum frame log-probability).  `model` is a CTCGreedySearch."""
    best, logp, hyp, hyp_len = model._search(xs_pad, xs_masks)
    lens = hyp_len.cpu().tolist()
    rows = hyp.cpu().tolist()
    return [r[:n] for r, n in zip(rows, lens)], logp.max(1).values
