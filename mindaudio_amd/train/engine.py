"""Training step of the Conformer CTC model on MI355X — what `TrainOneStepWithLossScaleCell(ASRModelWithAcc, Adam,
DynamicLossScaleUpdateCell)` does in the reference (mindaudio/utils/train_one_step.py:13-48,
examples/conformer/train.py:104-141, examples/conformer/asr_model.py:75-153), with a hand-written backward pass.

* All parameters live in ONE flat float32 buffer (masters), with a bf16 mirror at the same offsets (refreshed by one
  cast kernel per step) and bf16 transposed copies of the matmul weights (the dX GEMMs read those).  Gradients, Adam
  moments: flat buffers of the same layout -> one overflow check, one Adam launch, bucketed all-reduce on slices.
* Forward keeps what the backward needs on an explicit tape (no autograd graph); dropout masks are regenerated from
  (seed, site, index).  Matmuls are bf16 MFMA with float32 accumulation; residual stream, LayerNorm/BatchNorm
  statistics, softmax, CTC lattice, gradients of parameters and the optimizer are float32.
* Data parallel: one process per GPU; each rank runs the same step on its shard and the flat gradient is all-reduced
  (SUM, then 1/world inside the Adam scale) in per-layer buckets launched as soon as a layer's backward is done
  (torch.distributed: "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests of the bucketing logic).
"""
import math

import torch

from .. import _host, _lib, ops
from . import kernels as K
from . import kernels_x32 as X32

_PAD = 64  # every entry of the flat buffers starts on a 64-element boundary (16-byte aligned in bf16 and f32)


class FlatParams:
    """Name -> (offset, shape) views into flat float32 / bf16 buffers."""

    def __init__(self, entries, device, mirror_bf16=True):
        self.index = {}
        off = 0
        for name, shape in entries:
            n = 1
            for s in shape:
                n *= s
            self.index[name] = (off, tuple(shape), n)
            off += (n + _PAD - 1) // _PAD * _PAD
        self.size = off
        self.master = torch.zeros(off, dtype=torch.float32, device=device)
        # (one allocation: the flat gradient and, behind it, the step's overflow flag - one fill zeroes both)
        self._grad_alloc = torch.zeros(off + 64, dtype=torch.float32, device=device)
        self.grad = self._grad_alloc[:off]
        self.flag = self._grad_alloc[off:off + 1].view(torch.int32)
        self.exp_avg = torch.zeros(off, dtype=torch.float32, device=device)
        self.exp_avg_sq = torch.zeros(off, dtype=torch.float32, device=device)
        # matmul-operand copy of the masters: bf16 (throughput mode) or the masters themselves (float32 validation mode)
        self.bf16 = torch.zeros(off, dtype=torch.bfloat16, device=device) if mirror_bf16 else self.master
        self._pc, self._gc, self._wc = {}, {}, {}

    def _view(self, buf, name):
        off, shape, n = self.index[name]
        return buf[off:off + n].view(shape)

    # (the views are cached: the flat buffers never move, and a step asks for ~600 of them)
    def p(self, name):
        v = self._pc.get(name)
        if v is None:
            v = self._pc[name] = self._view(self.master, name)
        return v

    def g(self, name):
        v = self._gc.get(name)
        if v is None:
            v = self._gc[name] = self._view(self.grad, name)
        return v

    def w(self, name):
        v = self._wc.get(name)
        if v is None:
            v = self._wc[name] = self._view(self.bf16, name)
        return v

    def span(self, names):
        lo = min(self.index[n][0] for n in names)
        hi = max(self.index[n][0] + (self.index[n][2] + _PAD - 1) // _PAD * _PAD for n in names)
        return lo, hi


def asr_warmup_lr(step, base_lr=1e-3, warmup_steps=25000, start_steps=0):
    """ASRWarmupLR (mindaudio/scheduler/scheduler_factory.py:14-52): 0 at step 0, peak base_lr at warmup_steps."""
    s = float(step + start_steps)
    if s <= 0.0:
        return 0.0
    return base_lr * warmup_steps ** 0.5 * min(s ** -0.5, s * warmup_steps ** -1.5)


class DynamicLossScale:
    """DynamicLossScaleUpdateCell(loss_scale_value, scale_factor, scale_window) (train.py:126-129): halve (min 1) on
    overflow, double after `scale_window` consecutive clean steps."""

    def __init__(self, init=1024.0, factor=2.0, window=1000):
        self.scale, self.factor, self.window, self.good = float(init), float(factor), int(window), 0

    def update(self, overflow):
        if overflow:
            self.scale = max(self.scale / self.factor, 1.0)
            self.good = 0
        else:
            self.good += 1
            if self.good >= self.window:
                self.scale *= self.factor
                self.good = 0


class BucketedAllReduce:
    """Gradient all-reduce of the data-parallel step (the reference's grad_reducer, train_one_step.py:36): slices of the
    flat gradient buffer are summed across ranks as soon as the backward pass has finished them (asynchronous
    collectives on the backend's own stream), and `wait()` joins them before the optimizer.  With world == 1 it is a
    no-op unless `force_collective`: the collectives are then issued through the backend at world 1 as ReduceOp.AVG.  An in-place
    SUM over one rank is short-circuited by NCCL / RCCL without any device work; AVG is a pre-multiplied sum, for which the library
    launches its one-rank reduction KERNEL (x * 1/1: the identity, bit for bit) on its own stream - so the step must produce
    bit-identical masters, and does so only if RCCL's kernels beside the backward pass are ordered correctly against the
    library's launches on torch's current stream.  That is how the N > 1 concurrency is exercised on ONE GPU
    (tests/test_rccl_world1_gpu.py counts the RCCL kernels in a rocprofv3 trace).  Division by the world size happens inside the
    Adam kernel's scale."""

    def __init__(self, flat_grad, world_size=1, process_group=None, force_collective=False):
        self.grad, self.world, self.pg = flat_grad, int(world_size), process_group
        self.force = bool(force_collective)
        self.pending, self.launched = [], []

    def launch(self, lo, hi):
        self.launched.append((lo, hi))
        if self.world > 1 or self.force:
            import torch.distributed as dist

            op = dist.ReduceOp.AVG if (self.world == 1 and self.force) else dist.ReduceOp.SUM
            self.pending.append(dist.all_reduce(self.grad[lo:hi], op=op, group=self.pg, async_op=True))

    def wait(self):
        for work in self.pending:
            work.wait()
        covered = sorted(self.launched)
        self.pending, self.launched = [], []
        return covered


class _PoolEvent:
    __slots__ = ("ev",)

    def __init__(self):
        self.ev = torch.cuda.Event()

    def record_on(self, stream):
        self.ev.record(stream)
        return self.ev


_LAYER_W = (("ffm_w1", "feed_forward_macaron.w_1"), ("ffm_w2", "feed_forward_macaron.w_2"),
            ("o_w", "self_attn.linear_out"), ("ff_w1", "feed_forward.w_1"), ("ff_w2", "feed_forward.w_2"))
_LAYER_LN = ("norm_ff_macaron", "norm_mha", "norm_conv", "norm_ff", "norm_final")


def conformer_ctc_entries(d, hid, L, ks, heads, f2, V):
    """(name, shape) of every trainable tensor in flat order: front (subsampling + positional projections), blocks
    0..L-1, head (after_norm + CTC).  Matmul weights are stored in the layouts the kernels read (see _copy_params)."""
    ent = [("conv1_w", (d, 9)), ("conv1_b", (d,)), ("conv2_w", (d, 9 * d)), ("conv2_b", (d,)),
           ("out_w", (d, f2 * d)), ("out_b", (d,)), ("pos_w", (L * d, d))]
    for i in range(L):
        pre = "l%d." % i
        ent += [(pre + "ffm_w1", (hid, d)), (pre + "ffm_b1", (hid,)), (pre + "ffm_w2", (d, hid)), (pre + "ffm_b2", (d,)),
                (pre + "qkv_w", (3 * d, d)), (pre + "qkv_b", (3 * d,)), (pre + "o_w", (d, d)), (pre + "o_b", (d,)),
                (pre + "u", (heads, d // heads)), (pre + "v", (heads, d // heads)),
                (pre + "pw1_w", (2 * d, d)), (pre + "pw1_b", (2 * d,)), (pre + "dw_w", (d, ks)), (pre + "dw_b", (d,)),
                (pre + "bn_g", (d,)), (pre + "bn_b", (d,)), (pre + "pw2_w", (d, d)), (pre + "pw2_b", (d,)),
                (pre + "ff_w1", (hid, d)), (pre + "ff_b1", (hid,)), (pre + "ff_w2", (d, hid)), (pre + "ff_b2", (d,))]
        for ln in _LAYER_LN:
            ent += [(pre + ln + ".g", (d,)), (pre + ln + ".b", (d,))]
    ent += [("after_norm.g", (d,)), ("after_norm.b", (d,)), ("ctc_w", (V, d)), ("ctc_b", ((V + 63) // 64 * 64,))]
    return ent


_DEC_W = ("sa_qkv_w", "sa_o_w", "ca_q_w", "ca_kv_w", "ca_o_w", "ff_w1", "ff_w2")


def decoder_entries(d, hid, Ld, V):
    """Trainable tensors of the TransformerDecoder, appended after the encoder/CTC entries."""
    ent = [("dec.embed", (V, d))]
    # the memory-side projections (linear_k | linear_v of every layer's source attention, attention.py:86-157) of ALL layers lie next
    # to each other: they read the same encoder output, so the fused step runs them as ONE (Ld * 2d, d) product, their weight
    # gradient as one and the gradient w.r.t. the encoder output as one K = Ld * 2d product (round 6)
    ent += [("d%d.ca_kv_w" % i, (2 * d, d)) for i in range(Ld)]
    ent += [("d%d.ca_kv_b" % i, (2 * d,)) for i in range(Ld)]
    for i in range(Ld):
        pre = "d%d." % i
        ent += [(pre + "sa_qkv_w", (3 * d, d)), (pre + "sa_qkv_b", (3 * d,)), (pre + "sa_o_w", (d, d)), (pre + "sa_o_b", (d,)),
                (pre + "ca_q_w", (d, d)), (pre + "ca_q_b", (d,)),
                (pre + "ca_o_w", (d, d)), (pre + "ca_o_b", (d,)), (pre + "ff_w1", (hid, d)), (pre + "ff_b1", (hid,)),
                (pre + "ff_w2", (d, hid)), (pre + "ff_b2", (d,))]
        for ln in ("norm1", "norm2", "norm3"):
            ent += [(pre + ln + ".g", (d,)), (pre + ln + ".b", (d,))]
    ent += [("dec.after_norm.g", (d,)), ("dec.after_norm.b", (d,)), ("dec.out_w", (V, d)), ("dec.out_b", ((V + 63) // 64 * 64,))]
    return ent


def bucket_names(fp, L):
    layer_names = [[n for n in fp.index if n.startswith("l%d." % i)] for i in range(L)]
    return layer_names, ["conv1_w", "conv1_b", "conv2_w", "conv2_b", "out_w", "out_b"]


def bucket_spans(fp, L):
    """[(lo, hi)] of the gradient buckets in launch order: blocks L-1 .. 0, then the head and the front."""
    layer_names, embed_names = bucket_names(fp, L)
    spans = [fp.span(layer_names[li]) for li in reversed(range(L))]
    return spans + [fp.span(["after_norm.g", "ctc_b"]), fp.span(embed_names + ["pos_w"])]


# tools/wg_hunt.py, wg_twin.py, race_hunt.py, race_pairs.py ONLY (the two-hardware-queue corruption reproducer): keys
# "wg_stream" and "split_k_sums_on_second_stream".  Not an API: ConformerCTCTrainStep takes no such option.
_TWO_QUEUE_REPRODUCER = {}


class ConformerCTCTrainStep:
    """step(batch columns) -> (loss, cond, loss_scale, overflow, lr): one optimizer step of the ASR model.

    `model` is a mindaudio_amd.conformer.asr_model.ASRModel; its nn.Parameters provide the initial values (reference
    init, train.py:56 seeds) and receive the trained values back through `sync_to_module()`."""

    def __init__(self, model, base_lr=1e-3, warmup_steps=25000, loss_scale=1024.0, scale_factor=2.0, scale_window=1000,
                 beta1=0.9, beta2=0.999, eps=1e-8, dropout_rate=0.1, positional_dropout_rate=0.1, seed=777,
                 process_group=None, world_size=1, bn_momentum=0.1, rank=0, lr_step_rule="per_step", compute_type=None,
                 force_collective=False, fused=True, dw_group_blocks=6, own_stream="auto", start_steps=0, scheduler="warmuplr"):
        """start_steps: offset of the schedule index, `ASRWarmupLR(start_steps=start_epoch_num * steps_size)` of a resumed run
        (examples/conformer/train.py:117-124); scheduler: "warmuplr" (ASRWarmupLR) or "none" = Adam at the constant base_lr
        (train.py:126-127).
        compute_type: None / torch.bfloat16 = bf16 MFMA matmuls with float32 accumulation (the throughput mode);
        torch.float32 (the reference's default, mindaudio/models/conformer.py:61) = the float32 validation mode: every
        activation and product in float32 through the `_x32` kernels - same tape, same backward, same optimizer."""
        enc = model.encoder
        if compute_type not in (None, torch.bfloat16, torch.float32, "bfloat16", "float32"):
            raise ValueError("compute_type must be bfloat16 (default) or float32")
        self.x32 = compute_type in (torch.float32, "float32")
        # fused = the dense layers of a block run on fragment-packed weights with their element-wise neighbours (Swish + dropout,
        # residual + dropout + LayerNorm, Swish' + dropout, the next branch's dropout backward) in the launch's epilogue;
        # False = one launch per reference cell (what the float32 validation mode always runs)
        # The fused block launches (packed K = 256 dense layers, 256-wide LayerNorm epilogues) are built for the reference's
        # d_model = 256 configurations; d_model 512 / 768 / 1024 (64-wide heads; the reference's constructor takes any size,
        # models/conformer.py:293-313) run one launch per reference cell - the same policy as the evaluation forward (round 6;
        # until then this constructor refused them).
        self.fused = bool(fused) and not self.x32 and enc.d == 256
        # The step runs on ONE stream.  Rounds 3-4 also had a constructor option that put the weight-gradient products (and, in round 3,
        # the per-block batched sums) on a second stream: 0.4-14 % of fresh processes then saw a corrupted LayerNorm backward, narrowed
        # in round 4 to ONE kernel pair on two hardware queues (tn_reduce_batch_kernel beside layernorm_bwd_kernel) and
        # never explained.  The option is gone (VERDICT r4 #6c); the code path survives only for the reproducer scripts under tools/,
        # which set the module-level _TWO_QUEUE_REPRODUCER hook below before constructing an engine.
        self._wg_on = bool(_TWO_QUEUE_REPRODUCER.get("wg_stream")) and self.fused
        self._wg_split_ok = bool(_TWO_QUEUE_REPRODUCER.get("split_k_sums_on_second_stream"))
        self._wg, self._wg_keep, self._wg_pool, self._wg_next, self._wg_done, self._dw_par = None, [], [], 0, {}, 0
        self._wg_stream, self._wg_seen = None, {}
        self._wg_queue, self._main = [], None
        self.K, self.O = (X32, X32) if self.x32 else (K, ops)
        self.model, self.enc = model, enc
        self.dev = next(model.parameters()).device
        if self.dev.type == "cuda":
            _host.require_gpu()  # (+ the library's one-time kernel set-up, ma_init, before any stream of the step exists)
        self._wg_from = 2  # first step of a batch shape that may use the second stream (tools/wg_hunt.py sets 0)
        # dw_group_blocks = G > 0 (fused bf16 path, d_model and hidden multiples of 256): the eight weight-gradient products of a block
        # are not issued when the block's backward pass is done but together with those of the next G - 1 blocks, as ONE grid of
        # 256 x 256 tiles with the full contraction each (ma_gemm_tn_direct_group_bf16: 39 tiles per block, G = 6 fills the 256 CUs) -
        # no split-K partials, no reduction pass.  The gradient buckets of those G blocks go on the wire behind the group (L / G
        # all-reduce waves per step instead of L).  0 = one split-K grid + batched sum per block (round 3).
        self.dw_group_blocks = int(dw_group_blocks)
        # own_stream="auto": when gradients go through RCCL (world_size > 1 or force_collective) and the caller is on torch's DEFAULT
        # stream, the step runs on a stream the engine creates (the caller's stream waits for it at the end).  ROCm maps HIP streams
        # onto a few hardware queues and RCCL's stream was seen sharing the default stream's queue: its kernels then run in line with
        # the backward pass instead of beside it (profiles/r04_rccl_world1_trace.json).  False: always the caller's stream.
        self.own_stream, self._own = own_stream, None
        self._dq, self._dq_blocks, self._dq_dec = None, [], None
        # ln_bwd_fused: the four input-gradient products of a block that feed a LayerNorm backward (ff_w1 / ffm_w1 / pw1 / qkv
        # transposed, K = 2048 / 512 / 768) carry that LayerNorm backward and the next branch's dropout backward in their epilogue
        # (ma_gemm_rows_train_bf16 mode 5): 48 launches and 48 bf16 round trips of (M, 256) fewer per step
        self.ln_bwd_fused = True
        # ffn_one_launch: each feed-forward module's forward pass (w_1 + Swish + dropout -> u, h on the tape; w_2 + dropout + residual +
        # the LayerNorm (chain) behind it) is ONE launch (ma_ffn_train_bf16, the evaluation forward's hidden-slice-owner kernel with the
        # training work in its loop) instead of two: h is not read back for the second product
        self.ffn_one_launch = self.fused  # (and hidden % 256 == 0: settled below, once the model has been read)
        if self._wg_on and self.dev.type == "cuda":
            import ctypes

            self._wg_stream = torch.cuda.Stream(device=self.dev)
            self._wg_ptr = ctypes.c_void_p(self._wg_stream.cuda_stream)
        self.L = len(enc.encoders)
        self.d, self.heads = enc.d, enc.heads
        self.V = model.ctc.ctc_lo.out_features
        self.Vp = K.pad64(self.V)
        self.hidden = enc.encoders[0].feed_forward.w_1.out_features
        self.ffn_one_launch = self.ffn_one_launch and self.hidden % 256 == 0 and self.hidden <= 8192
        # ... and its backward pass as well (ma_ffn_train_bwd_bf16: dh -> du -> da -> the LayerNorm backward in front of the module);
        # the forward launch then leaves gk = swish'(.) * keep / (1 - p) on the tape in u's place
        self.ffn_bwd_one_launch = self.ffn_one_launch and self.ln_bwd_fused
        # ... and norm_final's backward of block l - 1 as a second stage of that launch's tail for block l's macaron module (the rows of
        # g it has just finished ARE norm_final's output gradient): 11 LayerNorm-backward launches and round trips of g fewer
        self.ln_final_chained = self.ffn_bwd_one_launch
        self._dw_direct = self.fused and self.dw_group_blocks > 0 and self.d % 256 == 0 and self.hidden % 256 == 0
        if self._dw_direct:
            # eight products per block in one direct group: keep a group inside the kernel's item table (a full group is also flushed
            # by DirectGroup.add itself - the decoder's 7 products per layer go into ONE group however many layers it has)
            cap = int(_lib.load().ma_gemm_tn_direct_max_items())
            self.dw_group_blocks = max(1, min(self.dw_group_blocks, cap // 8))
        # block_tables: from the third step of a batch shape on, a block's launches are issued by ONE C call each way
        # (ma_conformer_block_fwd_train / _bwd_train) from the argument table filled while the second step was walked from Python -
        # same calls, same buffers, same order; 4.2 -> ~1.5 ms of host time per step (train/block_table.py, csrc/block_table.hip)
        self.block_tables = self.fused and self._dw_direct
        self.block_table_min_sightings = 2        # a shape is recorded at its n-th step (>= 2), replayed from the next one on
        # device bytes all recorded tables together may pin.  A table keeps every buffer of its step (3-5 GB at the yaml's buckets);
        # the 16 buckets of conformer.yaml together are ~70 GB of the 288.  When the budget is reached the NEW shape stays walked
        # from Python (round 6; until then the oldest table was dropped - under the loader's round-robin over its buckets that is the
        # table needed next: every step evicted, re-walked and re-recorded a shape, 39 ms per hybrid step instead of 9.4,
        # tools/bucket_cycle_bench.py)
        self.block_table_max_bytes = 128 << 30
        self.block_table_max_blocks = 64          # kMaxBlocks of csrc/block_table.hip: encoder blocks + decoder layers + 3 segments
        self._table_bytes, self._table_warned, self._recording_tb, self._walk_shapes = {}, set(), None, set()
        if self._wg_on and not self._dw_direct and not self._wg_split_ok:
            raise ValueError("the two-queue reproducer needs the direct weight-gradient groups (dw_group_blocks > 0, d_model and hidden "
                             "multiples of 256) unless split_k_sums_on_second_stream is set as well (DESIGN 4.6.3)")
        self.ks = enc.kernel
        self.f2 = enc.embed.out.in_features // self.d
        self.p_drop, self.p_pos = float(dropout_rate), float(positional_dropout_rate)
        self.base_lr, self.warmup = base_lr, warmup_steps
        if scheduler not in ("warmuplr", "none"):
            raise ValueError("Only 'none', and 'warmuplr' are supported.")  # train.py:135
        self.scheduler, self.start_steps = scheduler, int(start_steps)
        self.b1, self.b2, self.eps = beta1, beta2, eps
        self.scaler = DynamicLossScale(loss_scale, scale_factor, scale_window)
        if lr_step_rule not in ("per_step", "mindspore23"):
            raise ValueError("lr_step_rule must be 'per_step' or 'mindspore23'")
        self.seed, self.global_step, self.applied_steps, self.calls = int(seed), 0, 0, 0
        self.lr_step_rule = lr_step_rule
        self.pg, self.world, self.rank = process_group, int(world_size), int(rank)
        self.bn_momentum = bn_momentum
        self.dec = getattr(model, "decoder", None)
        self.ctc_weight = float(model.ctc_weight)
        self.lsm = float(getattr(model, "lsm_weight", 0.0))
        self.len_norm = bool(getattr(model, "length_normalized_loss", False))
        if self.ctc_weight != 1.0 and self.dec is None:
            raise ValueError("ctc_weight != 1.0 needs model.decoder")
        self.Ld = len(self.dec.decoders) if self.dec is not None else 0
        self.dec_hidden = self.dec.decoders[0].feed_forward.w_1.out_features if self.dec is not None else 0
        self.last_acc = None
        # decoder_fused_launches: the TransformerDecoder's layers on the fused launches of the encoder blocks (dense + dropout + residual
        # + LayerNorm in one launch on packed weights; LayerNorm backward + the next dropout backward in one): 16 + 16 -> 10 + 12
        # launches per layer (round 6).  False: one launch per reference cell (what the float32 validation mode always runs).
        self.decoder_fused_launches = True
        self.decoder_embed_row_mask = True  # the embedding's backward skips the padded label rows (their gradient is exactly zero)
        self._dec_long_dw, self._dec_bucket_pending, self._dec_rows = True, False, 1
        self._build_flat()
        self.flag = self.fp.flag  # zeroed with the gradients at the start of forward_backward
        self.reducer = BucketedAllReduce(self.fp.grad, self.world, self.pg, force_collective)
        self.refresh_weights()

    # ---- flat parameter layout ------------------------------------------------------------------------------------
    def _build_flat(self):
        ent = conformer_ctc_entries(self.d, self.hidden, self.L, self.ks, self.heads, self.f2, self.V)
        if self.dec is not None:
            ent += decoder_entries(self.d, self.dec_hidden, self.Ld, self.V)
        self.fp = FlatParams(ent, self.dev, mirror_bf16=not self.x32)
        self._copy_params(to_flat=True)
        # BatchNorm running statistics (buffers, not optimised)
        self.bn_mean = [l.conv_module.norm.running_mean.detach().clone().float() for l in self.enc.encoders]
        self.bn_var = [l.conv_module.norm.running_var.detach().clone().float() for l in self.enc.encoders]
        self.layer_names, self.embed_names = bucket_names(self.fp, self.L)
        self.dec_names = [n for n in self.fp.index if n.startswith("dec.") or (n[0] == "d" and n[1].isdigit())]
        self._kv_all = None
        if self.dec is not None and self.Ld > 0:
            fp, d, Ld = self.fp, self.d, self.Ld
            ow, ob = fp.index["d0.ca_kv_w"][0], fp.index["d0.ca_kv_b"][0]
            contiguous = all(fp.index["d%d.ca_kv_w" % i][0] == ow + i * 2 * d * d and fp.index["d%d.ca_kv_b" % i][0] == ob + i * 2 * d
                             for i in range(Ld))
            if contiguous:  # (every tensor starts on a 64-element boundary: 2 d * d and 2 d are multiples of 64 for d = 256)
                n_w, n_b = Ld * 2 * d * d, Ld * 2 * d
                self._kv_all = dict(w=fp.bf16[ow:ow + n_w].view(Ld * 2 * d, d), b=fp.master[ob:ob + n_b],
                                    gw=fp.grad[ow:ow + n_w].view(Ld * 2 * d, d), gb=fp.grad[ob:ob + n_b])

    @torch.no_grad()
    def _copy_params(self, to_flat, grads_out=None):
        """to_flat: module parameters -> flat masters.  Otherwise flat -> module (masters into .data, or, when
        `grads_out` is a dict, the flat gradients into grads_out[parameter name] in the module's layouts)."""
        fp, e, d = self.fp, self.enc.embed, self.d
        buf = fp.grad if grads_out is not None else fp.master
        pname = {id(p): n for n, p in self.model.named_parameters()}

        class _Flat:  # fp.p(...) of the selected buffer
            @staticmethod
            def p(name):
                return fp._view(buf, name)

        def put(param, value):
            if grads_out is not None:
                grads_out[pname[id(param)]] = value.reshape(param.shape).clone()
            else:
                param.data.copy_(value.reshape(param.shape))

        def mv(name, param, fwd=None, bwd=None):
            if to_flat:
                src = param.detach().float()
                fp.p(name).copy_((fwd(src) if fwd else src).reshape(fp.p(name).shape))
            else:
                src = _Flat.p(name)
                put(param, bwd(src) if bwd else src)

        mv("conv1_w", e.conv1.weight)
        mv("conv1_b", e.conv1.bias)
        # (Cout, Cin, 3, 3) <-> (Cout, kh, kw, Cin): k = (kh, kw, c) of the implicit GEMM
        mv("conv2_w", e.conv2.weight, lambda w: w.permute(0, 2, 3, 1), lambda w: w.view(d, 3, 3, d).permute(0, 3, 1, 2))
        mv("conv2_b", e.conv2.bias)
        # reference flattens (c, f) (subsampling.py:76); the NHWC activation is (f, c)
        mv("out_w", e.out.weight, lambda w: w.view(d, d, self.f2).permute(0, 2, 1),
           lambda w: w.view(d, self.f2, d).permute(0, 2, 1))
        mv("out_b", e.out.bias)
        for i, l in enumerate(self.enc.encoders):
            pre = "l%d." % i
            a, cm = l.self_attn, l.conv_module
            if to_flat:
                fp.p("pos_w")[i * d:(i + 1) * d].copy_(a.linear_pos.weight.detach().float())
            else:
                put(a.linear_pos.weight, _Flat.p("pos_w")[i * d:(i + 1) * d])
            for short, path in _LAYER_W:
                mod = l
                for part in path.split("."):
                    mod = getattr(mod, part)
                mv(pre + short, mod.weight)
                mv(pre + short.replace("_w", "_b"), mod.bias)
            for j, lin in enumerate((a.linear_q, a.linear_k, a.linear_v)):
                if to_flat:
                    fp.p(pre + "qkv_w")[j * d:(j + 1) * d].copy_(lin.weight.detach().float())
                    fp.p(pre + "qkv_b")[j * d:(j + 1) * d].copy_(lin.bias.detach().float())
                else:
                    put(lin.weight, _Flat.p(pre + "qkv_w")[j * d:(j + 1) * d])
                    put(lin.bias, _Flat.p(pre + "qkv_b")[j * d:(j + 1) * d])
            mv(pre + "u", a.pos_bias_u)
            mv(pre + "v", a.pos_bias_v)
            mv(pre + "pw1_w", cm.pointwise_conv1.weight)
            mv(pre + "pw1_b", cm.pointwise_conv1.bias)
            mv(pre + "dw_w", cm.depthwise_conv.weight)
            mv(pre + "dw_b", cm.depthwise_conv.bias)
            mv(pre + "bn_g", cm.norm.weight)
            mv(pre + "bn_b", cm.norm.bias)
            mv(pre + "pw2_w", cm.pointwise_conv2.weight)
            mv(pre + "pw2_b", cm.pointwise_conv2.bias)
            for ln in _LAYER_LN:
                mv(pre + ln + ".g", getattr(l, ln).gamma)
                mv(pre + ln + ".b", getattr(l, ln).beta)
        mv("after_norm.g", self.enc.after_norm.gamma)
        mv("after_norm.b", self.enc.after_norm.beta)
        if self.dec is not None:
            dec = self.dec
            mv("dec.embed", dec.embed.weight)
            for i, l in enumerate(dec.decoders):
                pre = "d%d." % i
                sa, ca, ff = l.self_attn, l.src_attn, l.feed_forward
                for j, lin in enumerate((sa.linear_q, sa.linear_k, sa.linear_v)):
                    if to_flat:
                        fp.p(pre + "sa_qkv_w")[j * d:(j + 1) * d].copy_(lin.weight.detach().float())
                        fp.p(pre + "sa_qkv_b")[j * d:(j + 1) * d].copy_(lin.bias.detach().float())
                    else:
                        put(lin.weight, _Flat.p(pre + "sa_qkv_w")[j * d:(j + 1) * d])
                        put(lin.bias, _Flat.p(pre + "sa_qkv_b")[j * d:(j + 1) * d])
                for j, lin in enumerate((ca.linear_k, ca.linear_v)):
                    if to_flat:
                        fp.p(pre + "ca_kv_w")[j * d:(j + 1) * d].copy_(lin.weight.detach().float())
                        fp.p(pre + "ca_kv_b")[j * d:(j + 1) * d].copy_(lin.bias.detach().float())
                    else:
                        put(lin.weight, _Flat.p(pre + "ca_kv_w")[j * d:(j + 1) * d])
                        put(lin.bias, _Flat.p(pre + "ca_kv_b")[j * d:(j + 1) * d])
                mv(pre + "sa_o_w", sa.linear_out.weight)
                mv(pre + "sa_o_b", sa.linear_out.bias)
                mv(pre + "ca_q_w", ca.linear_q.weight)
                mv(pre + "ca_q_b", ca.linear_q.bias)
                mv(pre + "ca_o_w", ca.linear_out.weight)
                mv(pre + "ca_o_b", ca.linear_out.bias)
                mv(pre + "ff_w1", ff.w_1.weight)
                mv(pre + "ff_b1", ff.w_1.bias)
                mv(pre + "ff_w2", ff.w_2.weight)
                mv(pre + "ff_b2", ff.w_2.bias)
                for ln in ("norm1", "norm2", "norm3"):
                    mv(pre + ln + ".g", getattr(l, ln).gamma)
                    mv(pre + ln + ".b", getattr(l, ln).beta)
            mv("dec.after_norm.g", dec.after_norm.gamma)
            mv("dec.after_norm.b", dec.after_norm.beta)
            mv("dec.out_w", dec.output_layer.weight)
            if to_flat:
                fp.p("dec.out_b")[:self.V].copy_(dec.output_layer.bias.detach().float())
            else:
                put(dec.output_layer.bias, _Flat.p("dec.out_b")[:self.V])
        mv("ctc_w", self.model.ctc.ctc_lo.weight)
        if to_flat:
            fp.p("ctc_b")[:self.V].copy_(self.model.ctc.ctc_lo.bias.detach().float())
        else:
            put(self.model.ctc.ctc_lo.bias, _Flat.p("ctc_b")[:self.V])

    def gradients(self):
        """{parameter name: gradient} of the last forward_backward, in the module's (reference) layouts."""
        out = {}
        self._copy_params(to_flat=False, grads_out=out)
        return out

    def sync_to_module(self):
        """Write the trained masters (and BatchNorm running statistics) back into the nn.Module parameters."""
        self._copy_params(to_flat=False)
        for l, m, v in zip(self.enc.encoders, self.bn_mean, self.bn_var):
            l.conv_module.norm.running_mean.copy_(m)
            l.conv_module.norm.running_var.copy_(v)
        self.enc._prepared = None
        self.model.ctc._w = None
        if self.dec is not None:
            # (the evaluation forward's packed copies of the decoder's weights: left in place until the end of round 6, so an
            # evaluation between training steps - train.py's EvalCallback - ran the encoder of now with the decoder of the FIRST
            # evaluation; found by tools/recipe_learns.py --with-eval: the evaluation loss stopped at the untrained decoder's)
            self.dec._prepared = None

    @torch.no_grad()
    def sync_from_module(self, reset_moments=True):
        """The other direction: weights loaded into the nn.Module AFTER this engine was built (load_state_dict, a checkpoint
        importer) become the masters - the constructor reads the module once, and a later load left the engine training the old
        weights without a word.  BatchNorm running statistics come along; Adam's moments are zeroed unless `reset_moments` is False
        (the reference's checkpoints do not carry them either: a resumed run restarts them, train.py:117-133)."""
        self._copy_params(to_flat=True)
        for l, m, v in zip(self.enc.encoders, self.bn_mean, self.bn_var):
            m.copy_(l.conv_module.norm.running_mean.detach().float())
            v.copy_(l.conv_module.norm.running_var.detach().float())
        if reset_moments:
            self.fp.exp_avg.zero_()
            self.fp.exp_avg_sq.zero_()
        self.refresh_weights()

    @torch.no_grad()
    def refresh_weights(self, cast=True):
        """bf16 mirror of the masters (one cast launch; cast=False: the optimizer launch has written it) + transposed bf16 copies of the
        matmul weights."""
        self._sub_pk = None  # (the forward-only front end's packed copy of the two convolution weights)
        fp = self.fp
        if self.x32:  # the matmuls read the float32 masters themselves (dX = dY . W as an NN product: no transposed copies)
            return
        if cast:
            _lib.check(_lib.load().ma_cast_f32_bf16(fp.master.data_ptr(), fp.bf16.data_ptr(), fp.size,
                                                    torch.cuda.current_stream().cuda_stream), "cast")
        if not hasattr(self, "wt"):
            self.wt = {}
        names = ["conv2_w", "out_w", "ctc_w"]
        for i in range(self.L):
            names += ["l%d.%s" % (i, s) for s in ("ffm_w1", "ffm_w2", "qkv_w", "o_w", "pw1_w", "pw2_w", "ff_w1", "ff_w2")]
        alias = {}
        if self.dec is not None:
            names.append("dec.out_w")
            for i in range(self.Ld):
                names += ["d%d.%s" % (i, w) for w in _DEC_W]
            if getattr(self, "_kv_all", None) is not None and not self.x32:
                names.append("dec.ca_kv_all")  # (a view of the Ld contiguous ca_kv_w tensors, not an entry of the flat index)
                alias["dec.ca_kv_all"] = self._kv_all["w"]
        wv = lambda n: alias[n] if n in alias else fp.w(n)  # noqa: E731
        if getattr(self, "_wt_plan", None) is None:
            # one launch for all of them (ma_transpose_batch_bf16): the item list and the workgroup -> item map are built once, the
            # bf16 mirror and the transposed copies never move
            import ctypes

            import numpy as np

            items, block_item, first = [], [], 0
            for i, n in enumerate(names):
                w = wv(n)
                rows, cols = w.shape
                self.wt[n] = torch.zeros((cols, K.pad64(rows)), dtype=torch.bfloat16, device=self.dev)
                tr, tc = (rows + 63) // 64, (cols + 63) // 64
                ok = (w.stride(0) % 8 == 0 and self.wt[n].stride(0) % 8 == 0 and cols % 8 == 0 and w.data_ptr() % 16 == 0
                      and self.wt[n].data_ptr() % 16 == 0)
                if not ok:
                    items = None
                    break
                items.append(_lib.TransposeItem(w.data_ptr(), self.wt[n].data_ptr(), w.stride(0), self.wt[n].stride(0), rows, cols,
                                                first, tc))
                block_item += [i] * (tr * tc)
                first += tr * tc
            if items is None:
                self._wt_plan = False
            else:
                raw = (_lib.TransposeItem * len(items))(*items)
                dev_items = torch.from_numpy(np.frombuffer(bytes(raw), dtype=np.uint8).copy()).to(self.dev)
                dev_map = torch.tensor(block_item, dtype=torch.int32, device=self.dev)
                self._wt_plan = (dev_items, dev_map, first)
        if self._wt_plan:
            dev_items, dev_map, n_blocks = self._wt_plan
            _lib.check(_lib.load().ma_transpose_batch_bf16(dev_items.data_ptr(), dev_map.data_ptr(), n_blocks,
                                                           torch.cuda.current_stream().cuda_stream), "transpose_batch")
        else:
            for n in names:
                if n not in self.wt:
                    rows, cols = wv(n).shape
                    self.wt[n] = torch.zeros((cols, K.pad64(rows)), dtype=torch.bfloat16, device=self.dev)
                K.transpose(wv(n), out=self.wt[n])
        if self.fused:  # (after either transpose form: the packed copies are built from the mirror and the transposed copies)
            self._pack_weights()

    # fragment-packed copies of a block's dense weights (fused mode): name -> (source, kind) with kind 0 = K = 256 layers
    # (ma_gemm_k256_pack_bf16 layout), 1 = 256-output layers with a long contraction (ma_gemm_rows_pack_bf16 layout); ".t" sources
    # are the transposed copies (the input-gradient products)
    _PACKS = (("ffm_w1.k", "ffm_w1", False, 0), ("ffm_w2.r", "ffm_w2", False, 1), ("ffm_w2.tk", "ffm_w2", True, 0),
              ("ffm_w1.tr", "ffm_w1", True, 1), ("ff_w1.k", "ff_w1", False, 0), ("ff_w2.r", "ff_w2", False, 1),
              ("ff_w2.tk", "ff_w2", True, 0), ("ff_w1.tr", "ff_w1", True, 1), ("qkv_w.k", "qkv_w", False, 0),
              ("qkv_w.tr", "qkv_w", True, 1), ("o_w.k", "o_w", False, 0), ("o_w.tk", "o_w", True, 0), ("pw1_w.k", "pw1_w", False, 0),
              ("pw1_w.tr", "pw1_w", True, 1), ("pw2_w.k", "pw2_w", False, 0), ("pw2_w.tk", "pw2_w", True, 0))

    # the TransformerDecoder's dense layers (fused bf16 mode, round 6): same two layouts, names d<layer>.<key>
    _PACKS_DEC = (("sa_qkv_w.k", "sa_qkv_w", False, 0), ("sa_qkv_w.tr", "sa_qkv_w", True, 1), ("sa_o_w.k", "sa_o_w", False, 0),
                  ("sa_o_w.tk", "sa_o_w", True, 0), ("ca_q_w.k", "ca_q_w", False, 0), ("ca_q_w.tk", "ca_q_w", True, 0),
                  ("ca_o_w.k", "ca_o_w", False, 0), ("ca_o_w.tk", "ca_o_w", True, 0),
                  ("ff_w1.k", "ff_w1", False, 0), ("ff_w2.tk", "ff_w2", True, 0))

    _PACKS_FFN = (("ffm.f", "ffm_w1", False, 2), ("ffm.f", "ffm_w2", False, 3), ("ff.f", "ff_w1", False, 2), ("ff.f", "ff_w2", False, 3))
    _PACKS_FFN_T = (("ffm.ft", "ffm_w2", True, 2), ("ffm.ft", "ffm_w1", True, 3), ("ff.ft", "ff_w2", True, 2), ("ff.ft", "ff_w1", True, 3))

    @torch.no_grad()
    def _pack_weights(self):
        """One launch re-packs every block's dense weights (bf16 mirror and transposed copies -> MFMA-fragment order)."""
        lib = _lib.load()
        if getattr(self, "_pack_plan", None) is None:
            import numpy as np

            items, block_item, first, total = [], [], 0, 0
            self.pk = {}
            specs = []
            # the embed layer Linear(f2 * d -> d) (subsampling.py:46-47): 64 rows x 256 outputs per workgroup on a packed weight
            w = self.fp.w("out_w")
            pieces = int(lib.ma_pack_item_pieces(1, w.shape[0], w.shape[1]))
            if pieces > 0:
                specs.append(("out_w.r", w, w.shape[0], w.shape[1], 1, pieces, total))
                total += pieces * 16
            packs = self._PACKS
            if self.ffn_one_launch:  # the forward pass of the feed-forward modules reads the block format of W1 and W2 instead
                packs = tuple(pk for pk in packs if pk[0] not in ("ffm_w1.k", "ffm_w2.r", "ff_w1.k", "ff_w2.r")) + self._PACKS_FFN
            if self.ffn_bwd_one_launch:  # ... and their backward pass the block format of (W2^T, W1^T)
                packs = tuple(pk for pk in packs if pk[0] not in ("ffm_w2.tk", "ffm_w1.tr", "ff_w2.tk", "ff_w1.tr")) + self._PACKS_FFN_T
            for li in range(self.L):
                for key, src, transposed, kind in packs:
                    w = self.wt["l%d.%s" % (li, src)] if transposed else self.fp.w("l%d.%s" % (li, src))
                    n, k = w.shape[0], (self.fp.w("l%d.%s" % (li, src)).shape[0] if transposed else w.shape[1])
                    pieces = int(lib.ma_pack_item_pieces(kind, n, k))
                    _lib.check(min(pieces, 0), "pack %s" % key)
                    if kind == 3:  # the W2 half goes into the destination its W1 half (the previous spec) opened
                        specs.append(("l%d.%s" % (li, key), w, n, k, kind, pieces, specs[-1][6]))
                        continue
                    specs.append(("l%d.%s" % (li, key), w, n, k, kind, pieces, total))
                    total += pieces * 16 * (2 if kind == 2 else 1)
            self.dec_fused = False
            if self.dec is not None and self.d == 256 and self.dec_hidden % 256 == 0 and getattr(self, "_kv_all", None) is not None:
                dec_specs, ok = [], True
                w_all = self._kv_all["w"]
                pieces = int(lib.ma_pack_item_pieces(0, w_all.shape[0], w_all.shape[1]))
                if pieces <= 0:
                    ok = False
                dec_specs.append(("dec.ca_kv_all.k", w_all, w_all.shape[0], w_all.shape[1], 0, pieces, total))
                total += max(pieces, 0) * 16
                for li in range(self.Ld):
                    for key, src, transposed, kind in self._PACKS_DEC:
                        w = self.wt["d%d.%s" % (li, src)] if transposed else self.fp.w("d%d.%s" % (li, src))
                        n, k = w.shape[0], (self.fp.w("d%d.%s" % (li, src)).shape[0] if transposed else w.shape[1])
                        pieces = int(lib.ma_pack_item_pieces(kind, n, k))
                        if pieces <= 0:
                            ok = False
                        dec_specs.append(("d%d.%s" % (li, key), w, n, k, kind, pieces, total))
                        total += max(pieces, 0) * 16
                if ok:  # (a shape a pack kind does not cover: the decoder keeps its one-launch-per-cell walk)
                    specs += dec_specs
                    self.dec_fused = True
            arena = torch.empty(total, dtype=torch.uint8, device=self.dev)
            for name, w, n, k, kind, pieces, off in specs:
                self.pk[name] = arena[off:off + pieces * 16 * (2 if kind in (2, 3) else 1)]
                nblk = (pieces + 255) // 256
                items.append(_lib.PackItem(w.data_ptr(), arena.data_ptr() + off, w.stride(0), n, k, kind, first))
                block_item += [len(items) - 1] * nblk
                first += nblk
            raw = (_lib.PackItem * len(items))(*items)
            self._pack_plan = (torch.from_numpy(np.frombuffer(bytes(raw), dtype=np.uint8).copy()).to(self.dev),
                               torch.tensor(block_item, dtype=torch.int32, device=self.dev), first, arena)
        dev_items, dev_map, n_blocks, _ = self._pack_plan
        _lib.check(lib.ma_pack_batch_bf16(dev_items.data_ptr(), dev_map.data_ptr(), n_blocks,
                                          torch.cuda.current_stream().cuda_stream), "pack_batch")
        # the subsampling layer's second convolution runs on its own fragment order (conv2_packed.hip, as the evaluation forward)
        if getattr(self, "_conv2_pk", None) is None:
            self._conv2_pk = ops.conv2d_3x3s2_pack(self.fp.w("conv2_w").view(self.d, 3, 3, self.d))
        elif isinstance(self._conv2_pk, torch.Tensor):
            _lib.check(lib.ma_conv2d_3x3s2_pack_bf16(self.fp.w("conv2_w").data_ptr(), self.d, self.d, self._conv2_pk.data_ptr(),
                                                     torch.cuda.current_stream().cuda_stream), "conv2 pack")
        if self._conv2_pk is None:
            self._conv2_pk = False  # shape not covered: the general implicit GEMM

    # ---- helpers ---------------------------------------------------------------------------------------------------
    def _salt(self, layer, site):
        return (layer + 1) * 16 + site

    # Batch shapes whose plan - and block launch table: the tape and temporaries of one step, 2 - 4 GB - stay resident.  The reference
    # trains on static shapes (MindSpore graph mode): examples/conformer/conformer.yaml:71-72 has 16 (frame bucket, batch) pairs and pads
    # the labels to token_max_length, so 16 plans cover a whole epoch (~45 of 288 GB).
    _DW_PLANS_KEPT = 64  # (batch shapes whose plans - and launch tables - stay: the yaml has 16 buckets; least recently used first out)

    _DW_SUFFIXES = ("ffm_w1", "ffm_w2", "qkv_w", "o_w", "pw1_w", "pw2_w", "ff_w1", "ff_w2")

    _LN_SITES = ("norm_final", "norm_ff", "norm_conv", "norm_mha", "norm_ff_macaron")

    def _dw_plan_for(self, m):
        """Deferred parameter-gradient sums of a block (bf16 mode): the eight weight-gradient products write their split-K partials
        (and one partial bias-gradient vector per split) and the five LayerNorm backward launches their per-workgroup (dgamma | dbeta)
        partials into ONE arena that all blocks share, and ONE launch per block adds them into the flat gradient in a fixed order
        (ma_reduce_splits_batch_f32) instead of a 6 us reduction launch per product / per LayerNorm.  The arena is sized for the
        largest row count seen so far; the per-m offsets and device item tables are re-derived when m changes (one plan is kept)."""
        if self.x32:
            return None
        cur = self.__dict__.get("_dw_plan")
        if cur is not None and cur["m"] == m and cur.get("t2") == self._t2_cur:
            return cur
        # the last few shapes' plans are kept (bucketed batches come back: a plan carries the shape's block launch table), as long as
        # the shared arena they point into is the one they were derived for
        plans = self.__dict__.setdefault("_dw_plans", {})
        kept = plans.pop((m, self._t2_cur), None)
        if kept is not None and kept["arena"] is self.__dict__.get("_dw_arena"):
            plans[(m, self._t2_cur)] = kept  # (most recently used last)
            self._dw_plan = kept
            return kept
        import numpy as np

        lib, fp = _lib.load(), self.fp
        off, total = {}, 0
        for sfx in () if self._dw_direct else self._DW_SUFFIXES:  # (direct products write the flat gradient themselves)
            mo, no = fp.w("l0." + sfx).shape
            nbytes = int(lib.ma_gemm_tn_workspace_bytes(mo, no, m))
            off[sfx] = (total, nbytes, int(lib.ma_gemm_tn_splits(mo, no, m)))
            total += (nbytes + 255) // 256 * 256
        ln_parts = int(lib.ma_layernorm_bwd_parts(m))
        fused_parts = int(lib.ma_gemm_rows_train_parts(m))  # sites whose backward rides on an input-gradient product (ln_bwd_fused)
        ffn_parts = int(lib.ma_ffn_train_parts(m))          # ... or on the feed-forward module's one-launch backward
        most = max(ln_parts, fused_parts, ffn_parts)
        for site in self._LN_SITES:
            # (the region keeps the largest size; the item's split count is what the site's producer really writes)
            parts = fused_parts if (self.fused and self.ln_bwd_fused and site != "norm_final") else ln_parts
            if self.ffn_bwd_one_launch and site in ("norm_ff", "norm_ff_macaron"):
                parts = ffn_parts
            off[site] = (total, most * 512 * 4, parts)
            total += most * 512 * 4
        # the depthwise convolution's per-workgroup (d_dw_w | d_dw_b) partials (fused path)
        cm_parts = int(lib.ma_convmid_bwd_parts(m // self._t2_cur, self._t2_cur))
        cm_width = self.d * (self.ks + 1)
        off["convmid"] = (total, cm_parts * cm_width * 4, cm_parts)
        total += (cm_parts * cm_width * 4 + 255) // 256 * 256
        # the attention backward's workspace (its dpos / du / dv partials are reduced by the block's launch too) and the buffer the
        # positional projections' gradients of all blocks accumulate in
        import ctypes

        b_att = m // self._t2_cur
        att_bytes = int(lib.ma_relpos_attention_bwd_workspace_bytes(b_att, self._t2_cur, self.heads, self.d // self.heads))
        dp_off, bias_off, tp_, pph = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int32(), ctypes.c_int32()
        _lib.check(lib.ma_relpos_attention_bwd_layout(b_att, self._t2_cur, self.heads, self.d // self.heads, ctypes.byref(dp_off),
                                                      ctypes.byref(bias_off), ctypes.byref(tp_), ctypes.byref(pph)), "attention layout")
        total = (total + 255) // 256 * 256
        off["att_ws"] = (total, att_bytes, 0)
        total += (att_bytes + 255) // 256 * 256
        off["dpos_all"] = (total, self._t2_cur * self.L * self.d * 4, 0)
        total += (self._t2_cur * self.L * self.d * 4 + 255) // 256 * 256
        # Two copies of the arena, used by alternate blocks: with the weight-gradient products on their own stream (_wg) block li's
        # partials are still being summed there while the main stream's LayerNorm / attention backward of block li - 1 write theirs.
        half = (total + 255) // 256 * 256
        arena = self.__dict__.get("_dw_arena")
        if arena is None or arena.numel() < 2 * half:
            arena = self._dw_arena = torch.empty(2 * half, dtype=torch.uint8, device=self.dev)
        layers = []
        for li in range(self.L):
            items, block_item, first = [], [], 0
            base = arena.data_ptr() + (li & 1) * half

            def add(part_ptr, out, mn, ldo, n_cols, splits, pstride, tall, accumulate=True):
                nonlocal first
                nblk = (mn + 15) // 16 if tall else (mn + 1023) // 1024
                items.append(_lib.ReduceItem(part_ptr, out.data_ptr(), mn, ldo, n_cols, splits, 1.0,
                                             (1 if accumulate else 0) | (2 if tall else 0), first, pstride))
                block_item.extend([len(items) - 1] * nblk)
                first += nblk

            for sfx in () if self._dw_direct else self._DW_SUFFIXES:
                g = fp.g("l%d.%s" % (li, sfx))
                gb = fp.g("l%d.%s" % (li, sfx.replace("_w", "_b")))
                mo, no = g.shape
                o, nbytes, splits = off[sfx]
                add(base + o, g, mo * no, g.stride(0), no, splits, 0, False)
                add(base + o + splits * mo * no * 4, gb, mo, mo, mo, splits, 0, False)   # bias: partial column sums
            for site in self._LN_SITES:
                o, nbytes, parts = off[site]
                if site == "norm_final" and self._chain_final() and li < self.L - 1:
                    parts = ffn_parts  # written by block li + 1's macaron backward launch (ln_final_chained)
                gg = fp.g("l%d.%s.g" % (li, site))  # (g | b): 2 x 256 contiguous floats of the flat gradient
                assert fp.index["l%d.%s.b" % (li, site)][0] == fp.index["l%d.%s.g" % (li, site)][0] + 256
                add(base + o, gg, 512, 512, 512, parts, 512, True)
            if self.fused:
                o, nbytes, parts = off["convmid"]
                add(base + o, fp.g("l%d.dw_w" % li), self.d * self.ks, self.d * self.ks, self.d * self.ks, parts, cm_width, True)
                add(base + o + self.d * self.ks * 4, fp.g("l%d.dw_b" % li), self.d, self.d, self.d, parts, cm_width, True)
                t2, d, dk = self._t2_cur, self.d, self.d // self.heads
                ws_ptr = base + off["att_ws"][0]
                dpos_l = arena[off["dpos_all"][0]:off["dpos_all"][0] + off["dpos_all"][1]].view(torch.float32).view(t2, self.L * d)
                # dpos[t][c] of block li += sum over the batch of dp_part[b][t][c]
                # (stored, not accumulated: every block writes its own 256 columns once per step, so dpos_all needs no zero fill)
                # ("tall": one partial per utterance in flight per thread group - as a short item, 64 workgroups walked the 40
                # partials in five dependent rounds of HBM latency, the longest chain of the block's batched sum)
                add(ws_ptr + dp_off.value * 4, dpos_l[:, li * d:(li + 1) * d], t2 * d, self.L * d, d, b_att, tp_.value * d, True,
                    accumulate=False)
                for h in range(self.heads):  # du[h], dv[h] += sum over the (utterance, query block) partials of head h
                    hb = ws_ptr + (bias_off.value + h * pph.value * 128) * 4
                    add(hb, fp.g("l%d.u" % li)[h], dk, dk, dk, pph.value, 128, True)
                    add(hb + dk * 4, fp.g("l%d.v" % li)[h], dk, dk, dk, pph.value, 128, True)
            raw = (_lib.ReduceItem * len(items))(*items)
            layers.append((torch.from_numpy(np.frombuffer(bytes(raw), dtype=np.uint8).copy()).to(self.dev),
                           torch.tensor(block_item, dtype=torch.int32, device=self.dev), first))
        self._dw_plan = dict(m=m, t2=self._t2_cur, arena=arena, off=off, layers=layers, half=half)
        for k in [k for k, v in plans.items() if v["arena"] is not arena]:
            self._forget_plan(plans.pop(k))  # (the arena grew: the older plans' tables name the old one)
        plans[(m, self._t2_cur)] = self._dw_plan
        while len(plans) > self._DW_PLANS_KEPT:
            self._forget_plan(plans.pop(next(iter(plans))))
        if self.fused:
            o, nb, _ = off["att_ws"]
            self._dw_plan["att_ws"] = (arena[o:o + nb], arena[half + o:half + o + nb])
            o, nb, _ = off["dpos_all"]
            self._dw_plan["dpos_all"] = arena[o:o + nb].view(torch.float32).view(self._t2_cur, self.L * self.d)
        return self._dw_plan

    def _front_plan_for(self, m, t2):
        """The three weight gradients outside the blocks (CTC head, embed layer, positional projections): split-K partials into one
        arena, one batched sum at the end of the backward pass instead of five reduction launches (fused bf16 path)."""
        # (one plan per batch shape, the last _DW_PLANS_KEPT of them on one grow-only arena: with ONE cached plan every change of
        # bucket rebuilt it - an allocation and two synchronous host-to-device copies per step of a real epoch)
        plans = self.__dict__.setdefault("_front_plans", {})
        cur = plans.get((m, t2))
        if cur is not None:
            plans[(m, t2)] = plans.pop((m, t2))
            self._front_plan = cur
            return cur
        import numpy as np

        lib, fp = _lib.load(), self.fp
        prods = (("ctc", self.Vp, self.d, m, self.V, fp.g("ctc_w"), fp.g("ctc_b")),
                 ("out", self.d, self.f2 * self.d, m, self.d, fp.g("out_w"), fp.g("out_b")),
                 ("pos", self.L * self.d, self.d, t2, self.L * self.d, fp.g("pos_w"), None))
        off, total = {}, 0
        for name, mo, no, kc, mo_store, _, _ in prods:
            nbytes = int(lib.ma_gemm_tn_workspace_bytes(mo, no, kc))
            off[name] = (total, nbytes, int(lib.ma_gemm_tn_splits(mo, no, kc)))
            total += (nbytes + 255) // 256 * 256
        arena = self.__dict__.get("_front_arena")
        if arena is None or arena.numel() < total:
            arena = self._front_arena = torch.empty(max(total, int(1.25 * (arena.numel() if arena is not None else 0))),
                                                    dtype=torch.uint8, device=self.dev)
            plans.clear()  # (their item tables name the old arena)
        items, block_item, first = [], [], 0
        for name, mo, no, kc, mo_store, gw, gb in prods:
            o, _, splits = off[name]
            for part, out, mn, ldo, ncols in ((arena.data_ptr() + o, gw, mo_store * no, gw.stride(0), no),) + \
                    (((arena.data_ptr() + o + splits * mo_store * no * 4, gb, mo_store, mo_store, mo_store),) if gb is not None else ()):
                nblk = (mn + 1023) // 1024
                items.append(_lib.ReduceItem(part, out.data_ptr(), mn, ldo, ncols, splits, 1.0, 1, first, 0))
                block_item.extend([len(items) - 1] * nblk)
                first += nblk
        raw = (_lib.ReduceItem * len(items))(*items)
        self._front_plan = plans[(m, t2)] = dict(key=(m, t2), arena=arena, off=off, n_blocks=first,
                                                 items=torch.from_numpy(np.frombuffer(bytes(raw), dtype=np.uint8).copy()).to(self.dev),
                                                 block_item=torch.tensor(block_item, dtype=torch.int32, device=self.dev))
        while len(plans) > self._DW_PLANS_KEPT:
            plans.pop(next(iter(plans)))
        return self._front_plan

    def _front_dW(self, name, dy, x, rows_store=None, with_colsum=True):
        o, nbytes, _ = self._front_cur["off"][name]
        self.K.gemm_tn_partial(dy, x, self._front_cur["arena"][o:o + nbytes], with_colsum=with_colsum, rows_store=rows_store)

    def _block_table_for(self, b, t2, att_shape):
        """The launch table of the current batch shape: None (the blocks are walked from Python: first step of a shape, or a
        configuration the table does not cover), else a dict with `state` "record" (walk AND fill) or "replay", the table and the
        buffers at the blocks' boundary, which live at fixed addresses for as long as the table does."""
        plan = self._dw_cur
        if not (self.block_tables and self.fused and self._dw_direct and self._wg is None and plan is not None and not self.x32):
            return None
        key = (b, t2, tuple(att_shape), self.ffn_one_launch, self.ffn_bwd_one_launch, self.ln_final_chained,
               self.ln_bwd_fused, self.dw_group_blocks, self.p_drop, self.p_pos, self.bn_momentum, id(plan["arena"]),
               self.decoder_fused_launches, self.decoder_embed_row_mask)
        tables = plan.setdefault("tables", {})
        tb = plan["table"] = tables.get(key)  # (plan["table"]: the one in use, for tests and tools)
        if tb is None:
            while len(tables) >= 2:  # (e.g. the padding-mask and the chunk-mask form of one shape)
                self._drop_table(tables, next(iter(tables)))
            tb = plan["table"] = tables[key] = dict(key=key, state="seen", sightings=0)
        if key[:3] in self._walk_shapes:
            tb["state"] = "walk"
        if tb["state"] == "walk":  # a shape the table cannot cover (or could not be recorded for): walked from Python for good
            return None
        if tb["state"] == "seen":
            # the first step(s) of a shape warm the wrappers' pooled buffers and tell a recurring shape from a one-off (a loader that pads
            # to the batch maximum produces many shapes that never come back: recording each would pin 2-4 GB apiece)
            tb["sightings"] += 1
            if tb["sightings"] < max(2, int(self.block_table_min_sightings)):
                return None
            blocks = self.L + self.Ld + 3
            if blocks > int(self.block_table_max_blocks):
                self._table_warn("block tables: %d launch segments > the table's %d: this configuration is walked from Python"
                                 % (blocks, self.block_table_max_blocks))
                tb["state"] = "walk"
                return None
            from .block_table import BlockTable

            m, d, f32, bf = b * t2, self.d, torch.float32, torch.bfloat16
            tb.update(state="record", table=BlockTable(),
                      x_in=torch.empty((m, d), dtype=f32, device=self.dev), a_in=torch.empty((m, d), dtype=bf, device=self.dev),
                      mask_rows=torch.empty(m, dtype=f32, device=self.dev),
                      att_mask=torch.empty(att_shape, dtype=f32, device=self.dev),
                      pos_all=torch.empty((t2, self.L * d), dtype=bf, device=self.dev),
                      g=torch.empty((m, d), dtype=f32, device=self.dev))
            self._recording_tb = (tables, key)
            self._bn_snapshot = [(m_.clone(), v_.clone()) for m_, v_ in zip(self.bn_mean, self.bn_var)]
        return tb

    def _forget_plan(self, plan):
        for tb in plan.get("tables", {}).values():  # (the byte ledger must not keep a dropped plan's tables - and their buffers - alive)
            self._table_bytes.pop(id(tb), None)
            tb.clear()

    def _table_warn(self, msg):
        if msg not in self._table_warned:
            self._table_warned.add(msg)
            import warnings

            warnings.warn(msg, RuntimeWarning, stacklevel=3)

    def _drop_table(self, tables, key):
        tb = tables.pop(key, None)
        if tb is not None:
            self._table_bytes.pop(id(tb), None)
            tb.clear()

    def _table_recorded(self, tb):
        """End of a recorded step: the table replays from the next step on - unless the recorder met a call it cannot replay (then the
        shape is walked for good) or the byte budget of all tables is spent (then THIS shape stays walked, the recorded ones stay)."""
        table = tb["table"]
        self._recording_tb = None
        if table.broken is not None:
            self._table_warn("block tables: %s - this batch shape is walked from Python" % table.broken)
            for k in [k for k in tb if k not in ("key", "sightings")]:
                del tb[k]
            tb["state"] = "walk"
            return
        nbytes = table.nbytes()
        if self._table_bytes and sum(n for n, _ in self._table_bytes.values()) + nbytes > int(self.block_table_max_bytes):
            # over the budget: this shape is walked from Python from now on; the tables recorded so far stay
            self._table_warn("block tables: %.1f GB of launch tables are recorded (block_table_max_bytes); further batch shapes are "
                             "walked from Python" % (sum(n for n, _ in self._table_bytes.values()) / 2 ** 30))
            table.clear()
            for k in [k for k in tb if k not in ("key", "sightings")]:
                del tb[k]
            tb["state"] = "walk"
            return
        tb["state"] = "replay"
        self._table_bytes[id(tb)] = (nbytes, tb)

    def drop_block_tables(self):
        """Forget every recorded launch table (and the buffers they pin); the next steps of every shape are walked and re-recorded."""
        for plan in self._dw_plans.values() if hasattr(self, "_dw_plans") else ():
            for tb in (plan.pop("tables", None) or {}).values():
                table = tb.get("table")
                if table is not None and hasattr(table, "clear"):
                    table.clear()  # (the recorder's kept tensors: other references to the dict must not keep them pinned)
                tb.clear()
            plan.pop("table", None)
        self._table_bytes.clear()
        self._recording_tb = None

    def _chain_final(self):
        """norm_final's backward as the second stage of the block above's macaron backward launch (needs the one-launch forms)."""
        return self.ln_final_chained and self.ffn_bwd_one_launch and self.ffn_one_launch and self.fused

    def _ln_partials(self, site):
        """The arena slice that LayerNorm `site` of the current block writes its per-workgroup partials to (None: immediate sums)."""
        plan = getattr(self, "_dw_cur", None)
        if plan is None:
            return None
        o, nbytes, _ = plan["off"][site]
        o += self._dw_par * plan["half"]
        return plan["arena"][o:o + nbytes].view(torch.float32)

    def _dW(self, dy, x, wname, bname):
        """grad[wname] (N, K) += dy^T x ; grad[bname] += column sums of dy.  dy (M, N), x (M, K) bf16."""
        fp = self.fp
        plan = getattr(self, "_dw_cur", None)
        if self._dw_direct and wname[0] == "d" and wname[1].isdigit() and bname is not None and \
                self.K.gemm_tn_direct_ok(dy, x, fp.g(wname)):
            # a decoder layer's weight gradient: with the other 41 of the decoder in ONE grid at the end of its backward pass
            # (they were 43 split-K products + 88 reduction launches of ~10 us each: the hybrid step's decoder is launch-bound)
            if self._dec_long_dw and not self.block_tables and dy.shape[0] > 4 * self._dec_rows:
                # a decoder product over the ENCODER's rows (ca_kv_w: dkv^T memory, 10 200 rows against the decoder's 1 240): in the
                # decoder's grid its 2 tiles per layer ran 400 us with 244 CUs idle behind them (the grid lasts as long as its longest
                # tile).  It joins the encoder's first direct group instead - 234 + 12 tiles, still one resident round, every tile with a
                # full-length contraction; the decoder's gradient bucket goes on the wire behind that group (_dec_bucket_pending).
                # (Only with the launch tables off: a replayed encoder group holds the operands of the step it was recorded in - the
                # fused decoder walk keeps its one long operand at a fixed address for that, this per-layer form does not.)
                if self._dq is None:
                    self._dq = self.K.DirectGroup()
                self._dq.add(dy, x, fp.g(wname), fp.g(bname))
                self._dec_bucket_pending = True
                return
            if self._dq_dec is None:
                self._dq_dec = self.K.DirectGroup()
            self._dq_dec.add(dy, x, fp.g(wname), fp.g(bname))
            return
        if plan is not None and wname[0] == "l" and bname is not None:
            sfx = wname.split(".", 1)[1]
            if self._dw_direct and sfx in self._DW_SUFFIXES and self.K.gemm_tn_direct_ok(dy, x, fp.g(wname)):
                if self._dq is None:
                    self._dq = self.K.DirectGroup()
                self._dq.add(dy, x, fp.g(wname), fp.g(bname))  # issued with the group (_layer_done)
                return
            if sfx in plan["off"]:
                o, nbytes, _ = plan["off"][sfx]
                o += self._dw_par * plan["half"]
                if self.fused:
                    # queued: the block's eight products leave as ONE grouped launch when its backward pass is done (_layer_done) - on
                    # the weight-gradient stream behind one event pair, or on the main stream
                    self._wg_queue.append((dy, x, plan["arena"][o:o + nbytes]))
                    return
                self.K.gemm_tn_partial(dy, x, plan["arena"][o:o + nbytes], with_colsum=True)
                return
        self.K.gemm_tn(dy, x, fp.g(wname), colsum=fp.g(bname) if bname else None)

    def _dX(self, dy, wname, **kw):
        """dy (M, N) @ W (N, K): the input gradient of a dense layer whose weight is stored (out, in) like the reference's."""
        if self.x32:
            w = self.fp.w(wname)
            return X32.gemm_nn(dy[:, :w.shape[0]], w, residual=kw.get("residual"), out=kw.get("out"))
        return ops.gemm(dy, self.wt[wname], **kw)

    # ---- forward + backward ------------------------------------------------------------------------------------------
    @torch.no_grad()
    def forward_backward(self, xs_pad, ys_pad, xs_masks, ys_lengths, xs_chunk_masks=None, grad_scale=1.0, ys_in_pad=None,
                         ys_out_pad=None, ys_sub_masks=None, ys_masks=None):
        """Runs the training-mode forward and the backward pass; flat gradients hold grad_scale * dLoss/dparam.
        Returns the (unscaled) loss tensor."""
        args = (xs_pad, ys_pad, xs_masks, ys_lengths, xs_chunk_masks, grad_scale, ys_in_pad, ys_out_pad, ys_sub_masks, ys_masks)
        # A step that RECORDS a block launch table pins every buffer it allocates (2-4 GB per shape).  If that is what runs the device
        # out of memory: drop every table, give the memory back, walk this shape from Python for good and run the step again - with the
        # BatchNorm running statistics of before the attempt (the only state a forward pass changes besides the gradients).
        # (snapshot taken by _block_table_for when a shape enters its recording step)
        with _host.pinned_stream():
            oom_key = None
            try:
                return self._forward_backward(*args)
            except torch.OutOfMemoryError:
                # Only NOTE the failure here: while this block runs, the exception's traceback keeps the failed attempt's frames - its
                # tape, activations and the recording table's pinned buffers - alive, so nothing could be given back from in here
                # (the PyTorch FAQ's out-of-memory recovery pattern).
                rec = self._recording_tb
                if rec is None:
                    raise
                oom_key = rec[1]
            # (outside the handler: the traceback and with it every tensor of the failed attempt are gone)
            _lib.set_recording(None)
            self.drop_block_tables()
            self._dq = self._dq_dec = None
            self._dq_blocks.clear()
            self._wg_queue.clear()
            torch.cuda.synchronize(self.dev)
            torch.cuda.empty_cache()
            self._oom_freed_to = torch.cuda.memory_allocated(self.dev)  # (what the retry starts from; read by the tests)
            for (m0, v0), m, v in zip(self._bn_snapshot, self.bn_mean, self.bn_var):
                m.copy_(m0)
                v.copy_(v0)
            self._table_warn("block tables: out of memory while recording a batch shape - tables dropped, the shape is walked from Python")
            self._walk_shapes.add(oom_key[:3])
            return self._forward_backward(*args)

    def _encoder_forward(self, xs, xs_masks, xs_chunk_masks, seed, tables=True):
        """The encoder's training-mode forward (models/conformer.py:229-258 with self.training: dropout with the step's seed, BatchNorm
        batch statistics + running-statistics update): subsampling front end, positional dropout, the blocks, after_norm.  ONE
        implementation for the training step (_forward_backward) and for `ConformerEncoder.train()(xs, masks)`
        (encoder_forward_train); `tables` = the step's block launch tables and weight-gradient plans (training step only)."""
        fp, d, L = self.fp, self.d, self.L
        ops, K = self.O, self.K
        f32 = torch.float32
        pd, pp = self.p_drop, self.p_pos
        b = xs.shape[0]
        enc = self.enc
        act1 = None
        if not tables and self.fused and xs.shape[2] == 80 and getattr(self, "forward_only_fused_front", True):
            # nothing reads conv1's output again without a backward pass: both convolutions in one launch (subsample_fused.hip)
            if getattr(self, "_sub_pk", None) is None:
                self._sub_pk = ops.subsample_fused_pack(fp.p("conv1_w").reshape(d, 9).contiguous(), fp.w("conv2_w").view(d, 3, 3, d), 80)
            act2 = ops.subsample_fused(xs, self._sub_pk, fp.p("conv1_b"), fp.p("conv2_b"), enc.cmvn_mean, enc.cmvn_istd)
        else:
            act1 = ops.subsample_conv1(xs, fp.p("conv1_w"), fp.p("conv1_b"), enc.cmvn_mean, enc.cmvn_istd)
            if self.fused and isinstance(getattr(self, "_conv2_pk", None), torch.Tensor):
                act2 = ops.conv2d_3x3s2_packed(act1, self._conv2_pk, fp.p("conv2_b"), relu=True)
            else:
                act2 = ops.conv2d_3x3s2_nhwc(act1, fp.w("conv2_w").view(d, 3, 3, d), fp.p("conv2_b"), relu=True)
        _, t2, f2, c = act2.shape
        m = b * t2
        self._t2_cur = t2
        # (d_model 512 / 768 / 1024 - one launch per reference cell: every partial sum is taken where it is produced)
        self._dw_cur = self._dw_plan_for(m) if tables and self.d == 256 else None
        if xs_masks.shape[-1] != t2:
            raise ValueError("masks must be the subsampled pad mask (B, 1, %d), got %s" % (t2, tuple(xs_masks.shape)))
        chunked = xs_chunk_masks is not None and xs_chunk_masks.dim() == 3 and xs_chunk_masks.shape[1] == t2 and t2 > 1
        tb = self._block_table_for(b, t2, (b, t2, t2) if chunked else (b, t2)) if tables else None
        if tb is not None:  # (the blocks' inputs live at the table's addresses: conversion and placement in one launch)
            mask2d = tb["mask_rows"].view(b, t2).copy_(xs_masks.reshape(b, t2))
        else:
            mask2d = xs_masks.reshape(b, t2).to(f32).contiguous()
        mask_rows = mask2d.reshape(m)
        # (B, T') padding mask, or the (B, T', T') chunk masks of the streaming configuration (models/conformer.py:251-252)
        att_mask = enc._attention_mask(mask2d, xs_chunk_masks, b, t2, f32)
        if tb is not None and att_mask is not mask2d:
            att_mask = tb["att_mask"].copy_(att_mask)
        hlens = mask2d.sum(1).to(torch.int32)
        a2 = act2.view(m, f2 * c)
        if self.fused and "out_w.r" in self.pk:
            e = ops.gemm_rows_packed(a2, self.pk["out_w.r"].view(torch.bfloat16).view(d, -1), fp.p("out_b"), alpha=math.sqrt(d))
        else:
            e = ops.gemm(a2, fp.w("out_w"), bias=fp.p("out_b"), alpha=math.sqrt(d), out_dtype=f32)
        if pp > 0:
            x = K.dropout_add(None, e, 1.0, pp, seed, self._salt(-1, 0), out=tb["x_in"] if tb is not None else None)
        else:
            x = e if tb is None else tb["x_in"].copy_(e)
        pe = enc.pe[:t2].to(f32).contiguous()
        if pp > 0:
            pe = K.dropout_add(None, pe, 1.0, pp, seed, self._salt(-1, 1))
        pe_bf = ops.cast_bf16(pe)
        pos_all = ops.gemm(pe_bf, fp.w("pos_w"), out=tb["pos_all"] if tb is not None else None)  # (t2, L*256) bf16
        ctx_ = dict(seed=seed, b=b, t2=t2, m=m, mask_rows=mask_rows, att_mask=att_mask, pos_all=pos_all, table=tb, keep_tape=tables)
        if self.fused:
            x, enc_bf, tape = self._blocks_forward_fused(x, ctx_)
        else:
            x, enc_bf, tape = self._blocks_forward(x, ctx_)
        return dict(act1=act1, act2=act2, a2=a2, pe_bf=pe_bf, mask2d=mask2d, hlens=hlens, t2=t2, m=m, f2=f2, c=c, tb=tb, ctx=ctx_, x=x,
                    enc_bf=enc_bf, tape=tape)

    @torch.no_grad()
    def encoder_forward_train(self, xs_pad, xs_masks, xs_chunk_masks=None, seed=None):
        """Training-mode forward of the encoder alone on the engine's kernels: (B, T, idim) -> (B, T', d) float32.  The same launches
        as the training step's forward half (no launch tables, nothing kept for a backward pass); `seed` = the dropout seed (default:
        the step's seed rule at the current call counter, which it does not advance)."""
        if seed is None:
            seed = (self.seed + self.calls + 0x3c6ef35f * self.rank) & 0x7fffffff
        xs = xs_pad.to(torch.float32).contiguous()
        F = self._encoder_forward(xs, xs_masks, xs_chunk_masks, int(seed) & 0x7fffffff, tables=False)
        return F["x"].view(xs.shape[0], F["t2"], self.d)

    def _forward_backward(self, xs_pad, ys_pad, xs_masks, ys_lengths, xs_chunk_masks, grad_scale, ys_in_pad, ys_out_pad, ys_sub_masks,
                          ys_masks):
        fp, d, L = self.fp, self.d, self.L
        ops, K = self.O, self.K  # bf16 throughput kernels or their float32 validation twins
        f32, bf = torch.float32, torch.bfloat16
        # one dropout stream per (step, rank): data-parallel replicas must not draw the same masks
        seed = (self.seed + self.calls + 0x3c6ef35f * self.rank) & 0x7fffffff
        pd, pp = self.p_drop, self.p_pos
        b, t, idim = xs_pad.shape
        xs = xs_pad.to(f32).contiguous()
        enc = self.enc
        fp._grad_alloc.zero_()
        self._wg_next = 0
        self._wg_done.clear()
        self._wg_queue.clear()  # (a step that raised mid-backward must not leave stale products / pinned operands behind)
        self._wg_keep.clear()
        if self._dq is not None:
            self._dq.clear()
        if self._dq_dec is not None:
            self._dq_dec.clear()
        self._dq_blocks.clear()
        self._dec_bucket_pending = False
        # (experimental second stream: used from step _wg_from of a batch shape on - round 3's mitigation, kept; tools/wg_hunt.py sets 0)
        key = (b, t, idim)
        seen = self._wg_seen.get(key, 0)
        self._wg_seen[key] = seen + 1
        self._wg = self._wg_stream if (self._wg_stream is not None and seen >= self._wg_from) else None
        self._main = torch.cuda.current_stream() if self._wg is not None else None

        # ================= forward =================
        F = self._encoder_forward(xs, xs_masks, xs_chunk_masks, seed)
        act1, act2, a2, pe_bf, mask2d, hlens = F["act1"], F["act2"], F["a2"], F["pe_bf"], F["mask2d"], F["hlens"]
        t2, m, f2, c, tb, ctx_ = F["t2"], F["m"], F["f2"], F["c"], F["tb"], F["ctx"]
        x, enc_bf, tape = F["x"], F["enc_bf"], F["tape"]
        logits = torch.empty((m, self.Vp), dtype=f32, device=self.dev)
        ops.gemm(enc_bf, fp.w("ctc_w"), bias=fp.p("ctc_b"), out_dtype=f32, out=logits[:, :self.V])
        wc = self.ctc_weight
        loss, per_utt, dlog = K.ctc_loss_grad(logits, self.V, b, t2, ys_pad, hlens, ys_lengths, grad_scale * wc / b)
        d_mem = None
        if self.dec is not None:  # attention branch: loss = w * ctc + (1 - w) * att (asr_model.py:138-139)
            if ys_in_pad is None or ys_out_pad is None or ys_sub_masks is None or ys_masks is None:
                raise ValueError("the hybrid loss needs ys_in_pad, ys_out_pad, ys_sub_masks and ys_masks")
            # label_smoothing_loss.py:105-106: / batch, or / tokens (divided on the device) when length_normalized_loss
            loss_att, d_mem = self._decoder_forward_backward(enc_bf, mask2d, b, t2, ys_in_pad, ys_out_pad, ys_sub_masks,
                                                             ys_masks, grad_scale * (1.0 - wc) / (1.0 if self.len_norm else b),
                                                             seed, tb)
            self.last_loss_ctc, self.last_loss_att = loss, loss_att
            loss = wc * loss + (1.0 - wc) * loss_att

        # ================= backward =================
        # CTC head: logits = enc_bf W^T + b
        front = self.fused and dlog.shape[1] == self.Vp
        if front:
            self._front_cur = self._front_plan_for(m, t2)
            self._front_dW("ctc", dlog, enc_bf, rows_store=self.V)
        else:
            K.gemm_tn(dlog, enc_bf, fp.g("ctc_w"), colsum=fp.g("ctc_b"), rows_store=self.V)
        if d_mem is None:
            d_enc = self._dX(dlog, "ctc_w")            # (m, 256) bf16
        else:                                                   # + the decoder's gradient w.r.t. the encoder output
            d_enc = self._dX(dlog, "ctc_w", residual=d_mem, out_dtype=f32, out=d_mem)
        g = tb["g"] if tb is not None else torch.empty((m, d), dtype=f32, device=self.dev)
        K.layernorm_bwd(x, fp.p("after_norm.g"), d_enc, g, fp.g("after_norm.g"), fp.g("after_norm.b"), accumulate=False)
        if self.fused and self._dw_cur is not None:
            dpos_all = self._dw_cur["dpos_all"]  # (every element is written by the blocks' batched sums)
        else:
            dpos_all = torch.zeros((t2, L * d), dtype=f32, device=self.dev)
        if self.fused:
            self._blocks_backward_fused(g, tape, dpos_all, ctx_)
        else:
            self._blocks_backward(g, tape, dpos_all, ctx_)
        # positional projection of every layer: dW_pos (L*256, 256) = dpos_all^T pe
        if front:
            self._front_dW("pos", ops.cast_bf16(dpos_all), pe_bf, with_colsum=False)
        else:
            self._dW(ops.cast_bf16(dpos_all), pe_bf, "pos_w", None)
        # embedding: x = dropout(sqrt(d) * (a2 W_out^T + b))
        de = K.dropout_bwd(g, math.sqrt(d), pp, seed, self._salt(-1, 0))
        if front:
            self._front_dW("out", de, a2)
            fpl = self._front_cur  # the three sums in one launch
            _lib.check(_lib.load().ma_reduce_splits_batch_f32(fpl["items"].data_ptr(), fpl["block_item"].data_ptr(), fpl["n_blocks"],
                                                              _host.current_stream_ptr()), "front reduce")
        else:
            self._dW(de, a2, "out_w", "out_b")
        dact2 = self._dX(de, "out_w")                  # (m, f2*c) bf16
        K.relu_bwd(dact2, a2)
        dy2 = dact2.view(m * f2, c)
        if self._wg is not None:
            # the weight gradient of conv2 beside its input gradient and conv1's weight gradient (all three only need dy2 / act1)
            need = K.conv2d_dw_workspace_bytes(dy2.shape[0], c, c)
            if getattr(self, "_wg_ws", None) is None or self._wg_ws.numel() < need:
                self._wg_ws = torch.empty(need, dtype=torch.uint8, device=self.dev)
            self._wg.wait_event(self._wg_event().record_on(self._main))
            prev = _host.swap_pinned(self._wg_ptr)
            try:
                K.conv2d_dw(dy2, act1, fp.g("conv2_w"), fp.g("conv2_b"), ws=self._wg_ws)
            finally:
                _host.swap_pinned(prev)
            self._wg_keep.append((dy2, act1, dact2))
        else:
            K.conv2d_dw(dy2, act1, fp.g("conv2_w"), fp.g("conv2_b"))
        if self.fused:
            dact1 = K.conv2_dinput(dy2, self.wt["conv2_w"], act1)   # implicit GEMM per input-position parity class: no dcol
        else:
            dcol = self._dX(dy2, "conv2_w")            # (B*T2*F2, 9c) bf16
            dact1 = K.col2im_relu(dcol, act1)
        K.conv1_dw(dact1, xs, enc.cmvn_mean, enc.cmvn_istd, fp.g("conv1_w"), fp.g("conv1_b"))
        self._embed_done()
        return loss

    # ---- the blocks, one launch per reference cell ------------------------------------------------------------------------------
    def _blocks_forward(self, x, c):
        fp, d, L = self.fp, self.d, self.L
        ops, K = self.O, self.K
        f32 = torch.float32
        seed, b, t2, mask_rows, att_mask, pos_all, pd = c["seed"], c["b"], c["t2"], c["mask_rows"], c["att_mask"], c["pos_all"], self.p_drop
        tape = []
        for li in range(L):
            pre = "l%d." % li
            W, P = (lambda n, pre=pre: fp.w(pre + n)), (lambda n, pre=pre: fp.p(pre + n))
            T = {}
            # -- macaron FFN: x = x + 0.5 * drop(W2 drop(swish(W1 LN(x))))           models/conformer.py:109-112
            T["ffm"] = self._ffn_fwd(x, "ffm", "norm_ff_macaron", W, P, seed, li, 0)
            x = T["ffm"]["x_out"]
            # -- MHSA                                                                :117-135
            a = ops.layernorm(x, P("norm_mha.g"), P("norm_mha.b"))
            qkv = ops.gemm(a, W("qkv_w"), bias=P("qkv_b"))
            pos_l = pos_all[:, li * d:(li + 1) * d]
            ctx, lse = K.attention_fwd(qkv, pos_l, P("u"), P("v"), att_mask, b, t2, self.heads, d // self.heads)
            o = ops.gemm(ctx, W("o_w"), bias=P("o_b"))
            x_new = K.dropout_add(x, o, 1.0, pd, seed, self._salt(li, 2))
            T["mha"] = dict(x_in=x, a=a, qkv=qkv, ctx=ctx, lse=lse)
            x = x_new
            # -- convolution module                                                  :139-143, convolution.py:83-129
            a = ops.layernorm(x, P("norm_conv.g"), P("norm_conv.b"), row_scale=mask_rows)
            y = ops.gemm(a, W("pw1_w"), bias=P("pw1_b"))
            wv, z, stats = K.convmid_fwd_train(y, b, t2, P("dw_w"), P("dw_b"), P("bn_g"), P("bn_b"), self.bn_mean[li],
                                               self.bn_var[li], momentum=self.bn_momentum)
            o = ops.gemm(wv, W("pw2_w"), bias=P("pw2_b"), row_scale=mask_rows)
            x_new = K.dropout_add(x, o, 1.0, pd, seed, self._salt(li, 3))
            T["conv"] = dict(x_in=x, a=a, y=y, w=wv, z=z, stats=stats)
            x = x_new
            # -- FFN                                                                 :147-151
            T["ff"] = self._ffn_fwd(x, "ff", "norm_ff", W, P, seed, li, 6)
            x = T["ff"]["x_out"]
            # -- final LayerNorm of the block                                        :153-156
            T["final_in"] = x
            x = ops.layernorm(x, P("norm_final.g"), P("norm_final.b"), out_dtype=f32)
            tape.append(T)
        enc_bf = ops.layernorm(x, fp.p("after_norm.g"), fp.p("after_norm.b"))
        return x, enc_bf, tape

    def _blocks_backward(self, g, tape, dpos_all, c):
        fp, d, L = self.fp, self.d, self.L
        K = self.K
        seed, b, t2, mask_rows, att_mask, pos_all, pd = c["seed"], c["b"], c["t2"], c["mask_rows"], c["att_mask"], c["pos_all"], self.p_drop
        for li in reversed(range(L)):
            self._layer_begin(li)
            pre = "l%d." % li
            W, P, G = (lambda n, pre=pre: fp.w(pre + n)), (lambda n, pre=pre: fp.p(pre + n)), (lambda n, pre=pre: fp.g(pre + n))
            T = tape[li]
            K.layernorm_bwd(T["final_in"], P("norm_final.g"), g, g, G("norm_final.g"), G("norm_final.b"), accumulate=False,
                            partials=self._ln_partials("norm_final"))
            self._ffn_bwd(g, T["ff"], "ff", "norm_ff", pre, seed, li, 6)
            # conv module
            C = T["conv"]
            do = K.dropout_bwd(g, 1.0, pd, seed, self._salt(li, 3), row_scale=mask_rows)
            self._dW(do, C["w"], pre + "pw2_w", pre + "pw2_b")
            dwv = self._dX(do, pre + "pw2_w")
            dy = K.convmid_bwd(dwv, C["y"], C["z"], C["stats"], b, t2, P("dw_w"), P("bn_g"), P("bn_b"), G("dw_w"),
                               G("dw_b"), G("bn_g"), G("bn_b"))
            self._dW(dy, C["a"], pre + "pw1_w", pre + "pw1_b")
            da = self._dX(dy, pre + "pw1_w")
            K.layernorm_bwd(C["x_in"], P("norm_conv.g"), da, g, G("norm_conv.g"), G("norm_conv.b"), row_scale=mask_rows,
                            partials=self._ln_partials("norm_conv"))
            # MHSA
            A = T["mha"]
            do = K.dropout_bwd(g, 1.0, pd, seed, self._salt(li, 2))
            self._dW(do, A["ctx"], pre + "o_w", pre + "o_b")
            dctx = self._dX(do, pre + "o_w")
            dqkv = K.attention_bwd(A["qkv"], pos_all[:, li * d:(li + 1) * d], P("u"), P("v"), att_mask, A["ctx"], dctx,
                                   A["lse"], b, t2, dpos_all[:, li * d:(li + 1) * d], G("u"), G("v"), self.heads,
                                   d // self.heads)
            self._dW(dqkv, A["a"], pre + "qkv_w", pre + "qkv_b")
            da = self._dX(dqkv, pre + "qkv_w")
            K.layernorm_bwd(A["x_in"], P("norm_mha.g"), da, g, G("norm_mha.g"), G("norm_mha.b"),
                            partials=self._ln_partials("norm_mha"))
            self._ffn_bwd(g, T["ffm"], "ffm", "norm_ff_macaron", pre, seed, li, 0)
            self._layer_done(li)

    # ---- the blocks, fused: dense layers on fragment-packed weights with their element-wise neighbours in the epilogue ---------------
    def _blocks_forward_fused(self, x, c):
        """models/conformer.py:100-161 in training mode.  Per block 10 launches (23 un-fused; 12 in round 3): each feed-forward module
        is ONE launch (ffn_one_launch: w_1 + Swish + dropout -> the tape, w_2 + dropout + residual + the LayerNorm of the next cell; the
        two-launch form [w_1 ...] -> [w_2 ...] stays behind the switch), linear_q/k/v -> attention -> [linear_out + dropout + residual +
        norm_conv * mask], pointwise_conv1 -> depthwise / BatchNorm statistics / BatchNorm + Swish -> [pointwise_conv2 * mask + dropout +
        residual + norm_ff]; the last module of a block also applies norm_final and the LayerNorm that consumes it."""
        fp, d, L, K = self.fp, self.d, self.L, self.K
        seed, b, t2, mask_rows, att_mask, pos_all, pd = c["seed"], c["b"], c["t2"], c["mask_rows"], c["att_mask"], c["pos_all"], self.p_drop
        hid = self.hidden
        tb = c.get("table")
        a = ops.layernorm(x, fp.p("l0.norm_ff_macaron.g"), fp.p("l0.norm_ff_macaron.b"), out=tb["a_in"] if tb is not None else None)
        if tb is not None and tb["state"] == "replay":  # one C call per block: the launches recorded at this batch shape
            stream = _host.current_stream_ptr()
            for li in range(L):
                tb["table"].forward(li, seed, stream)
            return tb["out"]
        if tb is not None:
            if tb["table"].recorded:  # (a recorded step that raised before its backward pass was complete: start over)
                from .block_table import BlockTable

                tb["table"] = BlockTable()
            with tb["table"].recording(seed):
                tb["out"] = self._blocks_forward_walk(x, a, c, tb["table"])
            return tb["out"]
        return self._blocks_forward_walk(x, a, c, None)

    def _blocks_forward_walk(self, x, a, c, rec):
        fp, d, L, K = self.fp, self.d, self.L, self.K
        seed, b, t2, mask_rows, att_mask, pos_all, pd = c["seed"], c["b"], c["t2"], c["mask_rows"], c["att_mask"], c["pos_all"], self.p_drop
        hid = self.hidden
        tape, enc_bf = [], None
        keep = c.get("keep_tape", True)  # False: a forward that no backward pass follows (encoder_forward_train)
        for li in range(L):
            if rec is not None:
                rec.segment(False, li)
            pre = "l%d." % li
            P, PK = (lambda n, pre=pre: fp.p(pre + n)), (lambda n, pre=pre: self.pk[pre + n])
            ln = lambda n: (P(n + ".g"), P(n + ".b"))  # noqa: E731
            T = {}
            # -- macaron FFN
            if self.ffn_one_launch:
                u, h, x1, a1, _ = K.ffn_train(a, PK("ffm.f"), hid, P("ffm_b1"), pd, seed, self._salt(li, 0), P("ffm_b2"), x, 0.5, pd,
                                              self._salt(li, 1), ln1=ln("norm_mha"), tape_derivative=self.ffn_bwd_one_launch, tape=keep)
            else:
                u, h = K.dense_act_drop(a, PK("ffm_w1.k"), hid, P("ffm_b1"), pd, seed, self._salt(li, 0))
                x1, a1, _ = K.dense_join(h, PK("ffm_w2.r"), hid, P("ffm_b2"), x, 0.5, pd, seed, self._salt(li, 1), ln1=ln("norm_mha"))
            T["ffm"] = dict(x_in=x, a=a, u=u, h=h)
            # -- MHSA
            qkv = K.dense_plain(a1, PK("qkv_w.k"), 3 * d, d, bias=P("qkv_b"))
            ctx, lse = K.attention_fwd(qkv, pos_all[:, li * d:(li + 1) * d], P("u"), P("v"), att_mask, b, t2, self.heads, d // self.heads)
            x2, a2, _ = K.dense_join(ctx, PK("o_w.k"), d, P("o_b"), x1, 1.0, pd, seed, self._salt(li, 2), ln1=ln("norm_conv"),
                                     ln_row_scale=mask_rows)
            T["mha"] = dict(x_in=x1, a=a1, qkv=qkv, ctx=ctx, lse=lse)
            # -- convolution module
            y = K.dense_plain(a2, PK("pw1_w.k"), 2 * d, d, bias=P("pw1_b"))
            wv, z, stats = K.convmid_fwd_train(y, b, t2, P("dw_w"), P("dw_b"), P("bn_g"), P("bn_b"), self.bn_mean[li], self.bn_var[li],
                                               momentum=self.bn_momentum)
            x3, a3, _ = K.dense_join(wv, PK("pw2_w.k"), d, P("pw2_b"), x2, 1.0, pd, seed, self._salt(li, 3), row_scale=mask_rows,
                                     ln1=ln("norm_ff"))
            T["conv"] = dict(x_in=x2, a=a2, y=y, w=wv, z=z, stats=stats)
            # -- FFN, norm_final and the LayerNorm that reads its output (the next block's norm_ff_macaron, or after_norm)
            nxt = (fp.p("l%d.norm_ff_macaron.g" % (li + 1)), fp.p("l%d.norm_ff_macaron.b" % (li + 1))) if li + 1 < L else \
                (fp.p("after_norm.g"), fp.p("after_norm.b"))
            if self.ffn_one_launch:
                u, h, x4, a, x = K.ffn_train(a3, PK("ff.f"), hid, P("ff_b1"), pd, seed, self._salt(li, 6), P("ff_b2"), x3, 0.5, pd,
                                             self._salt(li, 7), ln1=ln("norm_final"), ln2=nxt, tape_derivative=self.ffn_bwd_one_launch,
                                             tape=keep)
            else:
                u, h = K.dense_act_drop(a3, PK("ff_w1.k"), hid, P("ff_b1"), pd, seed, self._salt(li, 6))
                x4, a, x = K.dense_join(h, PK("ff_w2.r"), hid, P("ff_b2"), x3, 0.5, pd, seed, self._salt(li, 7), ln1=ln("norm_final"),
                                        ln2=nxt)
            T["ff"] = dict(x_in=x3, a=a3, u=u, h=h)
            T["final_in"] = x4
            tape.append(T)
        return x, a, tape

    def _blocks_backward_fused(self, g, tape, dpos_all, c):
        """Per block 16 launches (40 un-fused; 28 in round 3): each feed-forward module's backward is ONE launch (ffn_bwd_one_launch:
        dh -> du -> da -> the LayerNorm backward in front of the module + the next branch's dropout backward), the other LayerNorm
        backwards ride on the input-gradient product that feeds them (ln_bwd_fused) or emit the dropout backward of the branch in front
        of them (bf16 dy), the input gradients run on packed transposed weights, and the weight gradients are queued for the direct
        groups (_dW)."""
        fp, d, L, K = self.fp, self.d, self.L, self.K
        seed, b, t2, mask_rows, att_mask, pos_all, pd = c["seed"], c["b"], c["t2"], c["mask_rows"], c["att_mask"], c["pos_all"], self.p_drop
        hid = self.hidden
        tb = c.get("table")
        if tb is not None and tb["state"] == "replay":
            stream = _host.current_stream_ptr()
            for li in reversed(range(L)):
                tb["table"].backward(li, seed, stream)
                # (what _layer_done / _flush_direct do besides launching: the finished groups' gradient buckets go on the wire)
                self._dq_blocks.append(li)
                if len(self._dq_blocks) >= self.dw_group_blocks or li == 0:
                    for blk in self._dq_blocks:
                        self.reducer.launch(*self.fp.span(self.layer_names[blk]))
                    self._launch_deferred_dec_bucket()
                    self._dq_blocks.clear()
            return
        if tb is not None:
            with tb["table"].recording(seed):
                self._blocks_backward_walk(g, tape, dpos_all, c, tb["table"])
            self._table_recorded(tb)
            return
        self._blocks_backward_walk(g, tape, dpos_all, c, None)

    def _blocks_backward_walk(self, g, tape, dpos_all, c, rec):
        fp, d, L, K = self.fp, self.d, self.L, self.K
        seed, b, t2, mask_rows, att_mask, pos_all, pd = c["seed"], c["b"], c["t2"], c["mask_rows"], c["att_mask"], c["pos_all"], self.p_drop
        hid = self.hidden
        chained_dy = None
        for li in reversed(range(L)):
            if rec is not None:
                rec.segment(True, li)
            self._layer_begin(li)
            pre = "l%d." % li
            P, G, PK = (lambda n, pre=pre: fp.p(pre + n)), (lambda n, pre=pre: fp.g(pre + n)), (lambda n, pre=pre: self.pk[pre + n])
            T = tape[li]

            def ffn_bwd(dy, F, key, ln, nxt, chain=None):
                self._dW(dy, F["h"], pre + key + "_w2", pre + key + "_b2")
                if self.ffn_bwd_one_launch:  # dh -> du -> da -> the LayerNorm backward: one launch (F["u"] holds gk)
                    du, dy_next = K.ffn_train_bwd(dy, PK(key + ".ft"), hid, F["u"], F["x_in"], P(ln + ".g"), g, self._ln_partials(ln),
                                                  nxt=nxt, chain=chain)
                    self._dW(du, F["a"], pre + key + "_w1", pre + key + "_b1")
                    return dy_next
                du = K.dense_act_drop_bwd(dy, PK(key + "_w2.tk"), hid, F["u"], pd, seed, self._salt(li, 0 if key == "ffm" else 6))
                self._dW(du, F["a"], pre + key + "_w1", pre + key + "_b1")
                if self.ln_bwd_fused:  # da = du . W_1 and the LayerNorm backward (+ the next branch's dropout backward): one launch
                    return K.dense_lnbwd(du, PK(key + "_w1.tr"), hid, F["x_in"], P(ln + ".g"), g, self._ln_partials(ln), nxt=nxt)
                da = K.dense_plain(du, PK(key + "_w1.tr"), d, hid)
                if nxt is None:
                    K.layernorm_bwd(F["x_in"], P(ln + ".g"), da, g, G(ln + ".g"), G(ln + ".b"), partials=self._ln_partials(ln))
                    return None
                return K.layernorm_bwd_next(F["x_in"], P(ln + ".g"), da, g, G(ln + ".g"), G(ln + ".b"), nxt,
                                            partials=self._ln_partials(ln))[1]

            if chained_dy is not None:  # norm_final's backward ran in block li + 1's macaron launch (ln_final_chained)
                dy, chained_dy = chained_dy, None
            else:
                _, dy = K.layernorm_bwd_next(T["final_in"], P("norm_final.g"), g, g, G("norm_final.g"), G("norm_final.b"),
                                             (0.5, pd, seed, self._salt(li, 7), None), accumulate=False,
                                             partials=self._ln_partials("norm_final"))
            do = ffn_bwd(dy, T["ff"], "ff", "norm_ff", (1.0, pd, seed, self._salt(li, 3), mask_rows))
            # conv module
            C = T["conv"]
            self._dW(do, C["w"], pre + "pw2_w", pre + "pw2_b")
            dwv = K.dense_plain(do, PK("pw2_w.tk"), d, d)
            dy = K.convmid_bwd(dwv, C["y"], C["z"], C["stats"], b, t2, P("dw_w"), P("bn_g"), P("bn_b"), G("dw_w"), G("dw_b"), G("bn_g"),
                               G("bn_b"), partials=self._ln_partials("convmid"))
            self._dW(dy, C["a"], pre + "pw1_w", pre + "pw1_b")
            if self.ln_bwd_fused:
                do = K.dense_lnbwd(dy, PK("pw1_w.tr"), 2 * d, C["x_in"], P("norm_conv.g"), g, self._ln_partials("norm_conv"),
                                   nxt=(1.0, pd, seed, self._salt(li, 2), None), row_scale=mask_rows)
            else:
                da = K.dense_plain(dy, PK("pw1_w.tr"), d, 2 * d)
                _, do = K.layernorm_bwd_next(C["x_in"], P("norm_conv.g"), da, g, G("norm_conv.g"), G("norm_conv.b"),
                                             (1.0, pd, seed, self._salt(li, 2), None), row_scale=mask_rows,
                                             partials=self._ln_partials("norm_conv"))
            # MHSA
            A = T["mha"]
            self._dW(do, A["ctx"], pre + "o_w", pre + "o_b")
            dctx = K.dense_plain(do, PK("o_w.tk"), d, d)
            # (dpos / du / dv: the partial sums stay in the plan's workspace and are added by the block's reduction launch)
            dqkv = K.attention_bwd(A["qkv"], pos_all[:, li * d:(li + 1) * d], P("u"), P("v"), att_mask, A["ctx"], dctx, A["lse"], b, t2,
                                   None, None, None, self.heads, d // self.heads, ws=self._dw_cur["att_ws"][self._dw_par])
            self._dW(dqkv, A["a"], pre + "qkv_w", pre + "qkv_b")
            if self.ln_bwd_fused:
                dy = K.dense_lnbwd(dqkv, PK("qkv_w.tr"), 3 * d, A["x_in"], P("norm_mha.g"), g, self._ln_partials("norm_mha"),
                                   nxt=(0.5, pd, seed, self._salt(li, 1), None))
            else:
                da = K.dense_plain(dqkv, PK("qkv_w.tr"), d, 3 * d)
                _, dy = K.layernorm_bwd_next(A["x_in"], P("norm_mha.g"), da, g, G("norm_mha.g"), G("norm_mha.b"),
                                             (0.5, pd, seed, self._salt(li, 1), None), partials=self._ln_partials("norm_mha"))
            if self._chain_final() and li > 0:
                # block li - 1's norm_final: its partials go to THAT block's half of the arena (block li + 1's sums, the last user of
                # that half, were launched on this stream long ago)
                prev = "l%d." % (li - 1)
                self._dw_par ^= 1
                part_prev = self._ln_partials("norm_final")
                self._dw_par ^= 1
                chained_dy = ffn_bwd(dy, T["ffm"], "ffm", "norm_ff_macaron", None,
                                     chain=(tape[li - 1]["final_in"], fp.p(prev + "norm_final.g"), part_prev,
                                            (0.5, pd, seed, self._salt(li - 1, 7), None)))
            else:
                ffn_bwd(dy, T["ffm"], "ffm", "norm_ff_macaron", None)
            self._layer_done(li)

    def _decoder_forward_backward(self, mem_bf, enc_mask2d, b, t2, ys_in_pad, ys_out_pad, ys_sub_masks, ys_masks, gscale,
                                  seed, tb=None):
        """TransformerDecoder forward + label-smoothing loss + backward (models/conformer.py:594-639,
        asr_model.py:154-186).  Fills the decoder gradients; returns (loss_att tensor, d_memory (B*T', 256) float32).
        With the encoder blocks' launch table `tb` (block_tables) the decoder's launches join it: decoder layer l is "block" L + l of
        the table (forward and backward), the embedding is the forward / backward of block L + Ld, the output layer the forward of
        block L + Ld + 1 and its backward that of block L + Ld + 2, with the loss (which takes the step's loss scale) issued between
        them; the label columns are copied into buffers the table owns."""
        dec, b_ = self.dec, b
        L1 = ys_in_pad.shape[1]
        md = b * L1
        f32 = torch.float32
        dtb = None
        # Does the ENCODER's backward run from its launch table in this step?  Its first direct group then already holds the product
        # the decoder's walk would queue (_decoder_walk_fused: the memory-side weight gradient) - also when the decoder itself is
        # walked from Python because this batch's label length is not the recorded one, which is the common case on real data.
        self._enc_replay_now = tb is not None and tb["state"] == "replay"
        if tb is not None and not self.len_norm and tb["state"] in ("record", "replay"):
            dtb = tb.get("dec")
            if tb["state"] == "record":
                dtb = tb["dec"] = dict(L1=L1, ls={},
                                       toks=torch.empty(md, dtype=torch.int32, device=self.dev),
                                       sub=torch.empty(tuple(ys_sub_masks.shape), dtype=f32, device=self.dev),
                                       pe=dec.pe[:L1].to(f32).contiguous().clone(),
                                       tgt=torch.empty(md, dtype=torch.int32, device=self.dev),
                                       tmask=torch.empty(md, dtype=f32, device=self.dev))
            elif dtb is not None and (dtb["L1"] != L1 or "out" not in dtb or tuple(dtb["sub"].shape) != tuple(ys_sub_masks.shape)):
                dtb = None  # another label length than the recorded step's: walked
        if dtb is None:
            toks = ys_in_pad.to(torch.int32).contiguous().reshape(-1)
            sub = ys_sub_masks.to(f32).contiguous()
            pe = dec.pe[:L1].to(f32).contiguous()
            tgt = ys_out_pad.to(torch.int32).contiguous().reshape(-1)
            tmask = ys_masks.to(f32).contiguous().reshape(-1)
            return self._decoder_walk(mem_bf, enc_mask2d, b_, t2, L1, toks, sub, pe, tgt, tmask, gscale, seed, None, None)
        # (conversion and placement in one launch each; the positional table is constant)
        toks, sub, pe = dtb["toks"].copy_(ys_in_pad.reshape(-1)), dtb["sub"].copy_(ys_sub_masks), dtb["pe"]
        tgt, tmask = dtb["tgt"].copy_(ys_out_pad.reshape(-1)), dtb["tmask"].copy_(ys_masks.reshape(-1))
        if tb["state"] == "record":
            with tb["table"].recording(seed):
                dtb["out"] = self._decoder_walk(mem_bf, enc_mask2d, b_, t2, L1, toks, sub, pe, tgt, tmask, gscale, seed, tb["table"],
                                                dtb["ls"])
            dtb["bucket_deferred"] = self._dec_bucket_pending  # (the replayed steps launch the bucket where the walked one did)
            stats, d_mem = dtb["out"]
        else:
            table, stream, L, Ld = tb["table"], _host.current_stream_ptr(), self.L, self.Ld
            table.forward(L + Ld, seed, stream)
            for li in range(Ld):
                table.forward(L + li, seed, stream)
            table.forward(L + Ld + 1, seed, stream)
            # (the loss carries the step's loss scale: issued from here, between the two halves of the output layer's segment pair)
            ls = dtb["ls"]
            self.K.label_smoothing_loss_grad(ls["logits"], self.V, tgt, tmask, self.lsm, gscale, bufs=ls)
            table.forward(L + Ld + 2, seed, stream)
            for li in reversed(range(Ld)):
                table.backward(L + li, seed, stream)
            table.backward(L + Ld, seed, stream)
            self._dec_bucket_pending = bool(dtb.get("bucket_deferred"))
            if self.dec_names and not self._dec_bucket_pending:
                self.reducer.launch(*self.fp.span(self.dec_names))
            stats, d_mem = dtb["out"]
        self.last_acc = stats[1] / stats[2]
        return stats[0] / b_, d_mem

    def _decoder_walk(self, mem_bf, enc_mask2d, b, t2, L1, toks, sub, pe, tgt, tmask, gscale, seed, rec, ls):
        """The decoder call by call; `rec`: the block table being filled (segments as _decoder_forward_backward lists them).
        Returns (loss_att, d_memory) - or (stats, d_memory) to the recording caller, which forms the loss from the kept `stats`."""
        fp, d, dec = self.fp, self.d, self.dec
        ops, K = self.O, self.K  # bf16 throughput kernels or their float32 validation twins
        tt = _host.torch()        # (allocations of a recorded step stay referenced by the table)
        f32 = torch.float32
        pd, pp = float(dec.dropout_rate), float(dec.positional_dropout_rate)
        eps = 1e-12  # models/conformer.py:417-419, 548
        dk = d // self.heads
        scale = 1.0 / dk  # q / sqrt(dk) . k / sqrt(dk) (attention.py:150-152)
        md, m = b * L1, b * t2
        emask = enc_mask2d
        RELU = _lib.ACT_RELU
        salt = lambda li, site: self._salt(100 + li, site)  # noqa: E731
        xscale = math.sqrt(d)
        seg = (lambda backward, blk: rec.segment(backward, blk)) if rec is not None else (lambda backward, blk: None)
        L, Ld = self.L, self.Ld
        self._dec_rows = md  # (what _dW compares a product's contraction length with)
        if self.fused and getattr(self, "dec_fused", False) and self.decoder_fused_launches:
            return self._decoder_walk_fused(mem_bf, emask, b, t2, L1, toks, sub, pe, tgt, tmask, gscale, seed, rec, ls, seg)
        seg(False, L + Ld)
        x = K.embed_posenc(toks, fp.p("dec.embed"), pe, L1, xscale, pp, seed, salt(-1, 0))
        tape = []
        for li in range(self.Ld):
            seg(False, L + li)
            pre = "d%d." % li
            W, P = (lambda n, pre=pre: fp.w(pre + n)), (lambda n, pre=pre: fp.p(pre + n))
            T = {"x0": x}
            a = ops.layernorm(x, P("norm1.g"), P("norm1.b"), eps=eps)
            qkv = ops.gemm(a, W("sa_qkv_w"), bias=P("sa_qkv_b"))
            ctx, probs = K.mha_small_fwd(qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:], sub, 2, b, L1, L1, scale, self.heads, dk)
            o = ops.gemm(ctx, W("sa_o_w"), bias=P("sa_o_b"))
            x1 = K.dropout_add(x, o, 1.0, pd, seed, salt(li, 0))
            T.update(a=a, qkv=qkv, ctx=ctx, probs=probs, x1=x1)
            a2 = ops.layernorm(x1, P("norm2.g"), P("norm2.b"), eps=eps)
            q = ops.gemm(a2, W("ca_q_w"), bias=P("ca_q_b"))
            kv = ops.gemm(mem_bf, W("ca_kv_w"), bias=P("ca_kv_b"))
            ctx2, probs2 = K.mha_small_fwd(q, kv[:, :d], kv[:, d:], emask, 1, b, L1, t2, scale, self.heads, dk)
            o2 = ops.gemm(ctx2, W("ca_o_w"), bias=P("ca_o_b"))
            x2 = K.dropout_add(x1, o2, 1.0, pd, seed, salt(li, 1))
            T.update(a2=a2, q=q, kv=kv, ctx2=ctx2, probs2=probs2, x2=x2)
            a3 = ops.layernorm(x2, P("norm3.g"), P("norm3.b"), eps=eps)
            u = ops.gemm(a3, W("ff_w1"), bias=P("ff_b1"))
            h = K.act_dropout_fwd(u, pd, seed, salt(li, 2), act=RELU)
            y = ops.gemm(h, W("ff_w2"), bias=P("ff_b2"))
            x = K.dropout_add(x2, y, 1.0, pd, seed, salt(li, 3))
            T.update(a3=a3, u=u, h=h)
            tape.append(T)
        seg(False, L + Ld + 1)
        yb = ops.layernorm(x, fp.p("dec.after_norm.g"), fp.p("dec.after_norm.b"), eps=eps)
        logits = tt.empty((md, self.Vp), dtype=f32, device=self.dev)
        ops.gemm(yb, fp.w("dec.out_w"), bias=fp.p("dec.out_b"), out_dtype=f32, out=logits[:, :self.V])
        seg(None, 0)  # (the loss takes the step's loss scale as an argument: never replayed with a recorded one)
        stats, dlog = K.label_smoothing_loss_grad(logits, self.V, tgt, tmask, self.lsm, gscale, normalize_length=self.len_norm,
                                                  **({} if ls is None else {"bufs": ls}))
        if ls is not None:
            ls["logits"] = logits
        # ---- backward ----
        seg(False, L + Ld + 2)
        K.gemm_tn(dlog, yb, fp.g("dec.out_w"), colsum=fp.g("dec.out_b"), rows_store=self.V)
        dy = self._dX(dlog, "dec.out_w")
        g = tt.empty((md, d), dtype=f32, device=self.dev)
        K.layernorm_bwd(x, fp.p("dec.after_norm.g"), dy, g, fp.g("dec.after_norm.g"), fp.g("dec.after_norm.b"),
                        accumulate=False, eps=eps)
        d_mem = tt.empty((m, d), dtype=f32, device=self.dev)  # (stored by the last layer's product, added to by the others: no fill)
        for li in reversed(range(self.Ld)):
            seg(True, L + li)
            pre = "d%d." % li
            P, G = (lambda n, pre=pre: fp.p(pre + n)), (lambda n, pre=pre: fp.g(pre + n))
            DX = lambda dy_, n, pre=pre, **kw: self._dX(dy_, pre + n, **kw)  # noqa: E731  (dy . W on the transposed copy)
            T = tape[li]
            # feed-forward
            dyf = K.dropout_bwd(g, 1.0, pd, seed, salt(li, 3))
            self._dW(dyf, T["h"], pre + "ff_w2", pre + "ff_b2")
            dh = DX(dyf, "ff_w2")
            du = K.act_dropout_bwd(T["u"], dh, pd, seed, salt(li, 2), out=dh, act=RELU)
            self._dW(du, T["a3"], pre + "ff_w1", pre + "ff_b1")
            K.layernorm_bwd(T["x2"], P("norm3.g"), DX(du, "ff_w1"), g, G("norm3.g"), G("norm3.b"), eps=eps)
            # source attention
            do = K.dropout_bwd(g, 1.0, pd, seed, salt(li, 1))
            self._dW(do, T["ctx2"], pre + "ca_o_w", pre + "ca_o_b")
            dctx = DX(do, "ca_o_w")
            dq = tt.empty_like(T["q"])
            dkv = tt.empty_like(T["kv"])
            K.mha_small_bwd(T["q"], T["kv"][:, :d], T["kv"][:, d:], T["probs2"], T["ctx2"], dctx, b, L1, t2, scale, dq,
                            dkv[:, :d], dkv[:, d:], self.heads, dk)
            self._dW(dq, T["a2"], pre + "ca_q_w", pre + "ca_q_b")
            K.layernorm_bwd(T["x1"], P("norm2.g"), DX(dq, "ca_q_w"), g, G("norm2.g"), G("norm2.b"), eps=eps)
            self._dW(dkv, mem_bf, pre + "ca_kv_w", pre + "ca_kv_b")
            DX(dkv, "ca_kv_w", residual=d_mem if li < self.Ld - 1 else None, out_dtype=f32, out=d_mem)
            # self attention
            do = K.dropout_bwd(g, 1.0, pd, seed, salt(li, 0))
            self._dW(do, T["ctx"], pre + "sa_o_w", pre + "sa_o_b")
            dctx = DX(do, "sa_o_w")
            dqkv = tt.empty_like(T["qkv"])
            qkv = T["qkv"]
            K.mha_small_bwd(qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:], T["probs"], T["ctx"], dctx, b, L1, L1, scale,
                            dqkv[:, :d], dqkv[:, d:2 * d], dqkv[:, 2 * d:], self.heads, dk)
            self._dW(dqkv, T["a"], pre + "sa_qkv_w", pre + "sa_qkv_b")
            K.layernorm_bwd(T["x0"], P("norm1.g"), DX(dqkv, "sa_qkv_w"), g, G("norm1.g"), G("norm1.b"), eps=eps)
        seg(True, L + Ld)
        K.embed_bwd(toks, g, fp.g("dec.embed"), xscale, pp, seed, salt(-1, 0))
        if self._dq_dec is not None:  # the decoder layers' weight gradients: one grid
            self._dq_dec.launch()
            self._dq_dec.clear()
        if rec is not None:
            rec.segment(None, 0)
            rec.keep.append(tape)
        if self.dec_names:
            self.reducer.launch(*fp.span(self.dec_names))
        if rec is not None:
            return stats, d_mem
        self.last_acc = stats[1] / stats[2]
        return stats[0] / (stats[2] if self.len_norm else b), d_mem

    def _dec_ln_plan(self, md):
        """The decoder's 3 Ld + 1 LayerNorm backwards leave their per-workgroup (dgamma | dbeta) partials in one arena and ONE batched
        launch adds them into the flat gradient at the end of the decoder's backward pass (19 reduction launches of 5.6 us fewer per
        hybrid step; the encoder blocks do the same per block, _dw_plan_for).
        One plan per row count md = B x (longest label + 1), which changes from batch to batch on real data: the plans of the last
        _DEC_LN_PLANS_KEPT row counts stay (building one costs two synchronous host-to-device copies - the item tables - i.e. the
        host's whole lead over the device), all on one grow-only arena; a launch table that recorded a plan keeps it alive itself."""
        plans = self.__dict__.setdefault("_dec_ln_plans", {})
        cur = plans.get(md)
        if cur is not None:
            plans[md] = plans.pop(md)  # (most recently used last)
            self._dec_ln = cur
            return cur
        import numpy as np

        fp = self.fp
        parts = int(_lib.load().ma_layernorm_bwd_parts(md))
        sites = ["dec.after_norm"] + ["d%d.%s" % (li, ln) for li in range(self.Ld) for ln in ("norm1", "norm2", "norm3")]
        need = len(sites) * parts * 512
        arena = self.__dict__.get("_dec_ln_arena")
        if arena is None or arena.numel() < need:
            # (a larger arena: the plans built on the old one go - recorded tables hold theirs, and through it the old arena)
            arena = self._dec_ln_arena = torch.empty(max(need, 2 * (arena.numel() if arena is not None else 0)), dtype=torch.float32,
                                                     device=self.dev)
            plans.clear()
        items, block_item, first, bufs = [], [], 0, {}
        for i, site in enumerate(sites):
            gg = fp.g(site + ".g")  # (g | b): 2 x 256 contiguous floats of the flat gradient
            assert fp.index[site + ".b"][0] == fp.index[site + ".g"][0] + 256
            bufs[site] = arena[i * parts * 512:(i + 1) * parts * 512]
            nblk = (512 + 15) // 16
            items.append(_lib.ReduceItem(bufs[site].data_ptr(), gg.data_ptr(), 512, 512, 512, parts, 1.0, 1 | 2, first, 512))
            block_item += [i] * nblk
            first += nblk
        raw = (_lib.ReduceItem * len(items))(*items)
        cur = dict(md=md, arena=arena, bufs=bufs, n_blocks=first,
                   items=torch.from_numpy(np.frombuffer(bytes(raw), dtype=np.uint8).copy()).to(self.dev),
                   block_item=torch.tensor(block_item, dtype=torch.int32, device=self.dev))
        plans[md] = cur
        while len(plans) > self._DEC_LN_PLANS_KEPT:
            plans.pop(next(iter(plans)))
        self._dec_ln = cur
        return cur

    _DEC_LN_PLANS_KEPT = 64

    def _decoder_walk_fused(self, mem_bf, emask, b, t2, L1, toks, sub, pe, tgt, tmask, gscale, seed, rec, ls, seg):
        """_decoder_walk on the fused launches the encoder blocks already use (round 6; bf16 mode, d_model 256): per layer 10 forward
        launches instead of 16 - every [dense + dropout + residual + the LayerNorm of the next cell] is ONE launch on a packed weight
        (K.dense_join), the other dense layers run on packed weights (K.dense_plain: the B x 31-token products are latency, not
        flops) - and 12 backward launches instead of 16: every LayerNorm backward emits the NEXT branch's dropout backward
        (K.layernorm_bwd_next), the input-gradient products run on the packed transposed weights.  Same tape, same dropout sites
        and salts, same rounding points as the walk above (the join rounds a W^T + b to bf16 before the dropout, as ops.gemm's bf16
        output does); the products sum in another order.  models/conformer.py:382-639, the hybrid step was 551 launches."""
        fp, d, dec = self.fp, self.d, self.dec
        K = self.K
        tt = _host.torch()
        f32 = torch.float32
        pd, pp = float(dec.dropout_rate), float(dec.positional_dropout_rate)
        eps = 1e-12
        dk = d // self.heads
        scale = 1.0 / dk
        md, m = b * L1, b * t2
        self._dec_rows = md
        hid = self.dec_hidden
        RELU = _lib.ACT_RELU
        salt = lambda li, site: self._salt(100 + li, site)  # noqa: E731
        xscale = math.sqrt(d)
        L, Ld = self.L, self.Ld
        seg(False, L + Ld)
        x = K.embed_posenc(toks, fp.p("dec.embed"), pe, L1, xscale, pp, seed, salt(-1, 0))
        a = ops.layernorm(x, fp.p("d0.norm1.g"), fp.p("d0.norm1.b"), eps=eps)
        # linear_k | linear_v of every layer's source attention on the encoder output: one (M, Ld * 2d) product (they were Ld launches
        # of 10 us over the same 10 200 rows)
        kv_all = K.dense_plain(mem_bf, self.pk["dec.ca_kv_all.k"], Ld * 2 * d, d, bias=self._kv_all["b"])
        tape = []
        for li in range(Ld):
            seg(False, L + li)
            pre = "d%d." % li
            P, PK = (lambda n, pre=pre: fp.p(pre + n)), (lambda n, pre=pre: self.pk[pre + n])
            T = {"x0": x, "a": a}
            qkv = K.dense_plain(a, PK("sa_qkv_w.k"), 3 * d, d, bias=P("sa_qkv_b"))
            ctx, probs = K.mha_small_fwd(qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:], sub, 2, b, L1, L1, scale, self.heads, dk)
            x1, a2, _ = K.dense_join(ctx, PK("sa_o_w.k"), d, P("sa_o_b"), x, 1.0, pd, seed, salt(li, 0),
                                     ln1=(P("norm2.g"), P("norm2.b")), eps=eps)
            T.update(qkv=qkv, ctx=ctx, probs=probs, x1=x1, a2=a2)
            q = K.dense_plain(a2, PK("ca_q_w.k"), d, d, bias=P("ca_q_b"))
            kv = kv_all[:, li * 2 * d:(li + 1) * 2 * d]
            ctx2, probs2 = K.mha_small_fwd(q, kv[:, :d], kv[:, d:], emask, 1, b, L1, t2, scale, self.heads, dk)
            x2, a3, _ = K.dense_join(ctx2, PK("ca_o_w.k"), d, P("ca_o_b"), x1, 1.0, pd, seed, salt(li, 1),
                                     ln1=(P("norm3.g"), P("norm3.b")), eps=eps)
            T.update(q=q, kv=kv, ctx2=ctx2, probs2=probs2, x2=x2, a3=a3)
            u, h = K.dense_act_drop(a3, PK("ff_w1.k"), hid, P("ff_b1"), pd, seed, salt(li, 2), act=RELU)  # w_1 + ReLU + dropout: one launch
            nxt = (fp.p("d%d.norm1.g" % (li + 1)), fp.p("d%d.norm1.b" % (li + 1))) if li + 1 < Ld else \
                (fp.p("dec.after_norm.g"), fp.p("dec.after_norm.b"))
            # (K = 2048 against 1 240 rows: on the row-owner kernel 26 workgroups each stream the whole weight - 30 us; the general GEMM's
            # 40 tiles walk 32 K-tiles each - 19 us + 5 + 5 for the join and the LayerNorm; split over K with the join and the LayerNorm
            # in the launch that sums the splits: two launches)
            x, a = K.dense_join_splitk(h, fp.w(pre + "ff_w2"), P("ff_b2"), x2, 1.0, pd, seed, salt(li, 3), ln1=nxt, eps=eps)
            T.update(u=u, h=h)
            tape.append(T)
        seg(False, L + Ld + 1)
        yb = a  # = LayerNorm(x; dec.after_norm), emitted by the last layer's join
        logits = tt.empty((md, self.Vp), dtype=f32, device=self.dev)
        ops.gemm(yb, fp.w("dec.out_w"), bias=fp.p("dec.out_b"), out_dtype=f32, out=logits[:, :self.V])
        seg(None, 0)  # (the loss takes the step's loss scale as an argument: never replayed with a recorded one)
        stats, dlog = K.label_smoothing_loss_grad(logits, self.V, tgt, tmask, self.lsm, gscale, normalize_length=self.len_norm,
                                                  **({} if ls is None else {"bufs": ls}))
        if ls is not None:
            ls["logits"] = logits
        # ---- backward ----
        seg(False, L + Ld + 2)
        K.gemm_tn(dlog, yb, fp.g("dec.out_w"), colsum=fp.g("dec.out_b"), rows_store=self.V)
        # (K = 4 288 against 1 240 rows: split over K like the feed-forward input gradient below - 32 -> ~12 us)
        dy = K.gemm_splitk(dlog, self.wt["dec.out_w"], tt.empty((md, d), dtype=f32, device=self.dev), accumulate=False)
        g = tt.empty((md, d), dtype=f32, device=self.dev)
        dlp = self._dec_ln_plan(md)  # (the LayerNorm backwards' partial sums: one batched reduction at the end)
        if rec is not None:
            # the recorded launches name this plan's arena and item table: they must outlive the plan of the next label width
            # (_dec_ln_plan keeps ONE; a batch with another longest transcript replaced it and the table's next replay faulted)
            rec.keep.append(dlp)
        lnp = dlp["bufs"]
        # after_norm's backward emits the dropout backward of the last layer's feed-forward join
        _, dyf = K.layernorm_bwd_next(x, fp.p("dec.after_norm.g"), dy, g, None, None, (1.0, pd, seed, salt(Ld - 1, 3), None),
                                      accumulate=False, eps=eps, partials=lnp["dec.after_norm"])
        d_mem = tt.empty((m, d), dtype=f32, device=self.dev)  # (stored by ONE product behind the layers: no fill, no accumulation)
        # dkv_all lives with the batch shape's plan, at ONE address for walked, recorded and replayed steps alike: the encoder's
        # first direct group reads it (below), and that group may be a recorded launch of another step's
        plan = getattr(self, "_dw_cur", None)
        merge = plan is not None and self._dw_direct and self._dec_long_dw
        if merge:
            dkv_all = plan.get("dkv_all")
            if dkv_all is None or tuple(dkv_all.shape) != (m, Ld * 2 * d):
                dkv_all = plan["dkv_all"] = torch.empty((m, Ld * 2 * d), dtype=torch.bfloat16, device=self.dev)
            merge = K.gemm_tn_direct_ok(dkv_all, mem_bf, self._kv_all["gw"])
        else:
            dkv_all = tt.empty((m, Ld * 2 * d), dtype=torch.bfloat16, device=self.dev)
        for li in reversed(range(Ld)):
            seg(True, L + li)
            pre = "d%d." % li
            P, G, PK = (lambda n, pre=pre: fp.p(pre + n)), (lambda n, pre=pre: fp.g(pre + n)), (lambda n, pre=pre: self.pk[pre + n])
            T = tape[li]
            # feed-forward
            self._dW(dyf, T["h"], pre + "ff_w2", pre + "ff_b2")
            du = K.dense_act_drop_bwd(dyf, PK("ff_w2.tk"), hid, T["u"], pd, seed, salt(li, 2), act=RELU)  # dh -> du: one launch
            self._dW(du, T["a3"], pre + "ff_w1", pre + "ff_b1")
            # (K = 2048 against 1 240 rows: 40 tiles of the general GEMM walk 32 K-tiles each - 17.5 us; split over K with partials in a
            # workspace and a fixed-order sum: two launches of ~5 us.  The LayerNorm backward reads the float32 sum as it is.)
            da = K.gemm_splitk(du, self.wt[pre + "ff_w1"], tt.empty((md, d), dtype=f32, device=self.dev), accumulate=False)
            _, do = K.layernorm_bwd_next(T["x2"], P("norm3.g"), da, g, None, None, (1.0, pd, seed, salt(li, 1), None), eps=eps,
                                         partials=lnp[pre + "norm3"])
            # source attention
            self._dW(do, T["ctx2"], pre + "ca_o_w", pre + "ca_o_b")
            dctx = K.dense_plain(do, PK("ca_o_w.tk"), d, d)
            dq = tt.empty_like(T["q"])
            dkv = dkv_all[:, li * 2 * d:(li + 1) * 2 * d]
            K.mha_small_bwd(T["q"], T["kv"][:, :d], T["kv"][:, d:], T["probs2"], T["ctx2"], dctx, b, L1, t2, scale, dq,
                            dkv[:, :d], dkv[:, d:], self.heads, dk)
            self._dW(dq, T["a2"], pre + "ca_q_w", pre + "ca_q_b")
            daq = K.dense_plain(dq, PK("ca_q_w.tk"), d, d)
            _, do = K.layernorm_bwd_next(T["x1"], P("norm2.g"), daq, g, None, None, (1.0, pd, seed, salt(li, 0), None), eps=eps,
                                         partials=lnp[pre + "norm2"])
            # self attention
            self._dW(do, T["ctx"], pre + "sa_o_w", pre + "sa_o_b")
            dctx = K.dense_plain(do, PK("sa_o_w.tk"), d, d)
            dqkv = tt.empty_like(T["qkv"])
            qkv = T["qkv"]
            K.mha_small_bwd(qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:], T["probs"], T["ctx"], dctx, b, L1, L1, scale,
                            dqkv[:, :d], dqkv[:, d:2 * d], dqkv[:, 2 * d:], self.heads, dk)
            self._dW(dqkv, T["a"], pre + "sa_qkv_w", pre + "sa_qkv_b")
            da = K.dense_plain(dqkv, PK("sa_qkv_w.tr"), d, 3 * d)
            if li > 0:  # ... and emits the dropout backward of the feed-forward join of the layer below
                _, dyf = K.layernorm_bwd_next(T["x0"], P("norm1.g"), da, g, None, None, (1.0, pd, seed, salt(li - 1, 3), None), eps=eps,
                                              partials=lnp[pre + "norm1"])
            else:
                K.layernorm_bwd(T["x0"], P("norm1.g"), da, g, None, None, eps=eps, partials=lnp[pre + "norm1"])
        seg(True, L + Ld)
        # the memory side of the source attentions, all layers at once: d_mem = dkv_all @ [ca_kv_w of every layer] (K = Ld * 2d), and
        # the Ld weight gradients as one (Ld * 2d, d) product in the encoder's first direct group (10 200 contraction rows: a
        # straggler beside the decoder's 1 240-row products, _dW)
        ops.gemm(dkv_all, self.wt["dec.ca_kv_all"], out_dtype=f32, out=d_mem)
        if merge:
            # (a replayed encoder backward has the item in its recorded group - same operands, same addresses; queueing it here as
            # well would leave it in a group nobody launches, with operands that die with this step)
            if not getattr(self, "_enc_replay_now", False):
                if self._dq is None:
                    self._dq = K.DirectGroup()
                self._dq.add(dkv_all, mem_bf, self._kv_all["gw"], self._kv_all["gb"])
            self._dec_bucket_pending = True
        else:
            K.gemm_tn(dkv_all, mem_bf, self._kv_all["gw"], colsum=self._kv_all["gb"])
        # (tmask: the padded label positions - their rows of g are exactly zero, the label mask keeps them out of every valid position)
        K.embed_bwd(toks, g, fp.g("dec.embed"), xscale, pp, seed, salt(-1, 0), row_keep=tmask if self.decoder_embed_row_mask else None)
        _lib.check(_lib.load().ma_reduce_splits_batch_f32(dlp["items"].data_ptr(), dlp["block_item"].data_ptr(), dlp["n_blocks"],
                                                          _host.current_stream_ptr()), "decoder LayerNorm sums")
        if self._dq_dec is not None:  # the decoder layers' weight gradients: one grid
            self._dq_dec.launch()
            self._dq_dec.clear()
        if rec is not None:
            rec.segment(None, 0)
            rec.keep.append(tape)
        if self.dec_names and not self._dec_bucket_pending:  # (pending: behind the encoder's first direct group, _flush_direct)
            self.reducer.launch(*fp.span(self.dec_names))
        if rec is not None:
            return stats, d_mem
        self.last_acc = stats[1] / stats[2]
        return stats[0] / (stats[2] if self.len_norm else b), d_mem

    def _ffn_fwd(self, x, key, ln, W, P, seed, li, s0):
        ops, K = self.O, self.K
        a = ops.layernorm(x, P(ln + ".g"), P(ln + ".b"))
        u = ops.gemm(a, W(key + "_w1"), bias=P(key + "_b1"))
        h = K.act_dropout_fwd(u, self.p_drop, seed, self._salt(li, s0))
        y = ops.gemm(h, W(key + "_w2"), bias=P(key + "_b2"))
        x_out = K.dropout_add(x, y, 0.5, self.p_drop, seed, self._salt(li, s0 + 1))
        return dict(x_in=x, a=a, u=u, h=h, x_out=x_out)

    def _ffn_bwd(self, g, T, key, ln, pre, seed, li, s0):
        fp, K = self.fp, self.K
        dy = K.dropout_bwd(g, 0.5, self.p_drop, seed, self._salt(li, s0 + 1))
        self._dW(dy, T["h"], pre + key + "_w2", pre + key + "_b2")
        dh = self._dX(dy, pre + key + "_w2")
        du = K.act_dropout_bwd(T["u"], dh, self.p_drop, seed, self._salt(li, s0), out=dh)
        self._dW(du, T["a"], pre + key + "_w1", pre + key + "_b1")
        da = self._dX(du, pre + key + "_w1")
        K.layernorm_bwd(T["x_in"], fp.p(pre + ln + ".g"), da, g, fp.g(pre + ln + ".g"), fp.g(pre + ln + ".b"),
                        partials=self._ln_partials(ln))

    # ---- data-parallel gradient reduction ----------------------------------------------------------------------------
    def _layer_done(self, li):
        """Backward of block li is complete: its slice of the flat gradient can go on the wire while earlier blocks
        are still being differentiated."""
        plan = getattr(self, "_dw_cur", None)
        if plan is not None:  # the split sums of this block's weight gradients, one launch
            items, block_item, n_blocks = plan["layers"][li]
            if self._dw_direct:
                # the block's partial sums (LayerNorm, depthwise convolution, attention biases) now; its weight-gradient products
                # leave with the group, and the group's gradient buckets behind them
                _lib.check(_lib.load().ma_reduce_splits_batch_f32(items.data_ptr(), block_item.data_ptr(), n_blocks,
                                                                  _host.current_stream_ptr()), "reduce_splits_batch")
                self._dq_blocks.append(li)
                if len(self._dq_blocks) >= self.dw_group_blocks or li == 0:
                    self._flush_direct()
                return
            if self._wg is not None:
                # behind the block's products on their stream AND the main stream's partials (LayerNorm / attention backward) and
                # direct sums (depthwise convolution, BatchNorm); the bucket goes on the wire behind the sums
                self._wg.wait_event(self._wg_event().record_on(self._main))
                prev = _host.swap_pinned(self._wg_ptr)
                try:
                    self.K.gemm_tn_partial_group(self._wg_queue, with_colsum=True)  # the block's eight products: one grid
                finally:
                    _host.swap_pinned(prev)
                # (dy / x stay referenced until the step's join: the caching allocator only orders reuse on the allocating stream)
                self._wg_keep.extend(self._wg_queue)
                self._wg_queue.clear()
                _lib.check(_lib.load().ma_reduce_splits_batch_f32(items.data_ptr(), block_item.data_ptr(), n_blocks,
                                                                  self._wg.cuda_stream), "reduce_splits_batch")
                if self.reducer.world > 1 or self.reducer.force:
                    with torch.cuda.stream(self._wg):
                        self.reducer.launch(*self.fp.span(self.layer_names[li]))
                else:
                    self.reducer.launch(*self.fp.span(self.layer_names[li]))
                done = self._wg_event()
                done.ev.record(self._wg)
                self._wg_done[li] = done
                if li == 0:  # the blocks are done: what follows on the main stream (dW_pos, the embed layer) reads / adds to dpos_all
                    self._main.wait_event(done.ev)  # and to gradients in the same flat buffer
                return
            if self._wg_queue:
                self.K.gemm_tn_partial_group(self._wg_queue, with_colsum=True)  # the block's eight products: one grid
                self._wg_queue.clear()
            _lib.check(_lib.load().ma_reduce_splits_batch_f32(items.data_ptr(), block_item.data_ptr(), n_blocks,
                                                              torch.cuda.current_stream().cuda_stream), "reduce_splits_batch")
        self.reducer.launch(*self.fp.span(self.layer_names[li]))

    def _flush_direct(self):
        if self._wg is not None:
            # the group's products (and its gradient buckets) on the second stream, behind what the main stream has produced so far;
            # the step's join (_embed_done) puts the optimizer behind them
            self._wg.wait_event(self._wg_event().record_on(self._main))
            prev = _host.swap_pinned(self._wg_ptr)
            try:
                if self._dq is not None:
                    self._dq.launch()
            finally:
                _host.swap_pinned(prev)
            if self._dq is not None:
                self._wg_keep.extend(self._dq.keep)  # (operands stay referenced until the join)
            if self.reducer.world > 1 or self.reducer.force:
                with torch.cuda.stream(self._wg):
                    for b in self._dq_blocks:
                        self.reducer.launch(*self.fp.span(self.layer_names[b]))
                    self._launch_deferred_dec_bucket()
            else:
                for b in self._dq_blocks:
                    self.reducer.launch(*self.fp.span(self.layer_names[b]))
                self._launch_deferred_dec_bucket()
        else:
            if self._dq is not None:
                self._dq.launch()
            for b in self._dq_blocks:
                self.reducer.launch(*self.fp.span(self.layer_names[b]))
            self._launch_deferred_dec_bucket()
        if self._dq is not None:
            self._dq.clear()
        self._dq_blocks.clear()

    def _launch_deferred_dec_bucket(self):
        """The decoder's gradient bucket when some of its weight gradients ride in the encoder's first direct group (_dW)."""
        if getattr(self, "_dec_bucket_pending", False):
            self._dec_bucket_pending = False
            if self.dec_names:
                self.reducer.launch(*self.fp.span(self.dec_names))

    def _layer_begin(self, li):
        """Backward of block li starts: its partial sums go to arena half li & 1, which block li + 2 used."""
        self._dw_par = li & 1
        if self._wg is not None:
            prev = self._wg_done.pop(li + 2, None)
            if prev is not None:
                self._main.wait_event(prev.ev)

    def _wg_event(self):
        """An event from the step's pool (events are re-recorded every step: creating one costs more than recording it)."""
        if self._wg_next == len(self._wg_pool):
            self._wg_pool.append(_PoolEvent())
        e = self._wg_pool[self._wg_next]
        self._wg_next += 1
        return e

    def _embed_done(self):
        self.reducer.launch(*self.fp.span(["after_norm.g", "ctc_b"]))
        if self._wg is not None:
            # the front bucket holds conv2's weight gradient (second stream) and the main stream's sums: behind both; then the main
            # stream continues behind everything the second stream has done in this step
            self._wg.wait_event(self._wg_event().record_on(self._main))
            if self.reducer.world > 1 or self.reducer.force:
                with torch.cuda.stream(self._wg):
                    self.reducer.launch(*self.fp.span(self.embed_names + ["pos_w"]))
            else:
                self.reducer.launch(*self.fp.span(self.embed_names + ["pos_w"]))
            done = self._wg_event()
            done.ev.record(self._wg)
            self._main.wait_event(done.ev)
            self._wg_keep.clear()
            return
        self.reducer.launch(*self.fp.span(self.embed_names + ["pos_w"]))

    # ---- one optimizer step ------------------------------------------------------------------------------------------
    @torch.no_grad()
    def step(self, xs_pad, ys_pad, ys_in_pad=None, ys_out_pad=None, r_ys_in_pad=None, r_ys_out_pad=None,
             xs_masks=None, ys_sub_masks=None, ys_masks=None, ys_lengths=None, xs_chunk_masks=None):
        """Same 11 inputs as ASRModelWithAcc.construct (train.py:38-50).  Returns (loss, cond, scaling_sens, overflow, lr),
        the five fields of TrainOneStepWithLossScaleCell.construct (train_one_step.py:48); cond == overflow (bool).

        Counters (restated from train_one_step.py:41-46 and MindSpore's optimizer semantics, which are third-party and NOT
        in the reference tree, hence unpinned):
        * `global_step` indexes the LR schedule.  `self.optimizer.get_lr()` is called on every step, overflow or not
          (train_one_step.py:44), so with lr_step_rule="per_step" (default) it advances by one per step() call.
          lr_step_rule="mindspore23" additionally restates that MindSpore >= 2.0's `Optimizer.get_lr()` itself increments
          global_step and `nn.Adam.construct` calls it a second time when the update is applied: the returned lr is
          schedule(global_step), the update uses schedule(global_step + 1), and a clean step advances the counter by two.
        * `applied_steps` counts the updates actually applied: Adam's beta1_power / beta2_power live inside the optimizer,
          which is not executed on overflow (train_one_step.py:45-46), so the bias correction uses this counter."""
        loss, scale, lr = self.enqueue_step(xs_pad, ys_pad, ys_in_pad, ys_out_pad, r_ys_in_pad, r_ys_out_pad, xs_masks, ys_sub_masks,
                                            ys_masks, ys_lengths, xs_chunk_masks)
        return self.finish_step(loss, scale, lr)

    @torch.no_grad()
    def enqueue_step(self, xs_pad, ys_pad, ys_in_pad=None, ys_out_pad=None, r_ys_in_pad=None, r_ys_out_pad=None,
                     xs_masks=None, ys_sub_masks=None, ys_masks=None, ys_lengths=None, xs_chunk_masks=None):
        """Everything of step() that only ENQUEUES device work (forward, backward, all-reduce launches, overflow check, Adam, the bf16
        weight refresh); no host read-back.  finish_step() reads the overflow flag and moves the host-side counters.  (bench.py
        times this call behind a busy GPU: `train_dp.host_enqueue_ms`.)"""
        if self.own_stream == "auto" and self.dev.type == "cuda" and (self.world > 1 or self.reducer.force):
            cur = torch.cuda.current_stream(self.dev)
            if cur == torch.cuda.default_stream(self.dev):
                if self._own is None:
                    self._own = torch.cuda.Stream(device=self.dev)
                self._own.wait_stream(cur)
                with torch.cuda.stream(self._own):
                    out = self._enqueue_step(xs_pad, ys_pad, ys_in_pad, ys_out_pad, xs_masks, ys_sub_masks, ys_masks, ys_lengths,
                                             xs_chunk_masks)
                cur.wait_stream(self._own)
                return out
        return self._enqueue_step(xs_pad, ys_pad, ys_in_pad, ys_out_pad, xs_masks, ys_sub_masks, ys_masks, ys_lengths, xs_chunk_masks)

    def _enqueue_step(self, xs_pad, ys_pad, ys_in_pad, ys_out_pad, xs_masks, ys_sub_masks, ys_masks, ys_lengths, xs_chunk_masks):
        scale = self.scaler.scale
        loss = self.forward_backward(xs_pad, ys_pad, xs_masks, ys_lengths, xs_chunk_masks, grad_scale=scale,
                                     ys_in_pad=ys_in_pad, ys_out_pad=ys_out_pad, ys_sub_masks=ys_sub_masks,
                                     ys_masks=ys_masks)
        K = self.K
        self.reducer.wait()
        K.grad_overflow(self.fp.grad, self.flag)
        lr = self.lr_at(self.global_step)
        two = self.lr_step_rule == "mindspore23"
        lr_opt = self.lr_at(self.global_step + 1) if two else lr
        tstep = self.applied_steps + 1
        lr_t = lr_opt * math.sqrt(1.0 - self.b2 ** tstep) / (1.0 - self.b1 ** tstep)
        mirrored = K.adam(self.fp.master, self.fp.grad, self.fp.exp_avg, self.fp.exp_avg_sq, lr_t, self.b1, self.b2, self.eps,
                          1.0 / (scale * self.world), self.flag, **({} if self.x32 else {"mirror": self.fp.bf16}))
        self.refresh_weights(cast=not mirrored)  # (the bf16 mirror of the masters left with the update)
        # What finish_step hands to the host - the overflow flag and the loss - leaves for pinned memory here, with an event behind it:
        # finish_step waits for THAT, not for the stream, so whatever the caller enqueues between enqueue_step and finish_step (the next
        # batch's feature and collate launches, conformer/train.py) runs on the device while the host already reads this step's results.
        host = self.__dict__.get("_step_host")
        if host is None:
            host = self._step_host = (torch.empty(1, dtype=torch.float32, pin_memory=True), torch.empty(1, dtype=self.flag.dtype, pin_memory=True),
                                      torch.cuda.Event())
        host[0].copy_(loss.detach().reshape(1).to(torch.float32), non_blocking=True)
        host[1].copy_(self.flag.reshape(1), non_blocking=True)
        host[2].record()
        return loss, scale, lr

    def lr_at(self, step):
        """Learning rate of schedule index `step`: ASRWarmupLR at step + start_steps, or the constant of `scheduler: none`."""
        if self.scheduler == "none":
            return float(self.base_lr)
        return asr_warmup_lr(step, self.base_lr, self.warmup, self.start_steps)

    def finish_step(self, loss, scale, lr):
        two = self.lr_step_rule == "mindspore23"
        host = self.__dict__.get("_step_host")
        if host is not None:
            host[2].synchronize()
            overflow = bool(int(host[1][0]))  # the reference also hands `cond` back to the host every step
            loss = torch.tensor(float(host[0][0]))  # (the step's loss as a host scalar: the device tensor stays in last_loss_*)
        else:
            overflow = bool(int(self.flag.item()))
        self.scaler.update(overflow)
        self.calls += 1
        self.global_step += 1
        if not overflow:
            self.applied_steps += 1
            if two:
                self.global_step += 1
        return loss, overflow, scale, overflow, lr
