// CTC branch of the ASR model, forward (loss value) — mindaudio/loss/ctc_loss.py:53-64:
//   ys_hat = Dense(256 -> V) [ma_gemm_bf16] -> float32 log_softmax over V -> CTCLossV2(blank = 0, reduction none,
//   zero_infinity) -> sum over the batch / B.
// The (T, B, V) log-prob tensor is never materialised: ctc_lse_kernel reduces each logits row to its log-sum-exp,
// ctc_alpha_kernel runs the alpha recursion of one utterance per wave (extended label states on lanes, neighbours
// fetched with wave shuffles), reading only logit[t, blank] and logit[t, label_s].
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include <type_traits>

#include "../../include/mindaudio_amd.h"

#include "launch.h"

namespace ma {

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
  return v;
}
__device__ __forceinline__ float wave_add(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
__device__ __forceinline__ float log_add(float a, float b) {
  const float m = fmaxf(a, b);
  if (m == -INFINITY) return -INFINITY;
  return m + log1pf(expf(fminf(a, b) - m));
}

// log(e^a + e^b [+ e^c]) of the recursions in one pass: three independent hardware exponentials and one logarithm (v_exp_f32 /
// v_log_f32, ~1 ulp) instead of two dependent libm log1p(exp()) chains - the recursion is ONE wave per utterance, so a time step
// costs the latency of its dependent instructions (0.85 us per step before).  log(sum) lies in [0, ln 3]: the absolute error per
// step is ~1e-7, against losses of 1e2 .. 1e3.
__device__ __forceinline__ float log_add3(float a, float b, float c, bool use_c) {
  const float cc = use_c ? c : -INFINITY;
  const float m = fmaxf(fmaxf(a, b), cc);
  if (m == -INFINITY) return -INFINITY;
  const float sum = __expf(a - m) + __expf(b - m) + __expf(cc - m);
  return m + __logf(sum);
}

// one wave per row: lse[row] = log(sum_v exp(logits[row, v]))
// Rows of up to kLseRegQ * 256 columns with 16-byte alignment (the training step: V = 4233 in rows of 4288 floats) are read ONCE, as
// float4 per lane into registers - maximum, then the sum of exponentials from the registers (round 4; as two passes of 4-byte loads
// the kernel ran at 2.1 TB/s of the 175 MB it reads).  Columns [V, 4 ceil(V / 4)) lie inside the row (ld % 4 == 0, ld >= V) and are
// masked by index, whatever they hold.
constexpr int kLseRegQ = 20;
__global__ __launch_bounds__(256) void ctc_lse_kernel(const float* __restrict__ logits, int64_t ld, int64_t rows, int V,
                                                      float* __restrict__ lse) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* p = logits + row * ld;
  const int nq = (V + 3) >> 2;
  if ((ld & 3) == 0 && (reinterpret_cast<uintptr_t>(logits) & 15) == 0 && nq <= kLseRegQ * 64) {
    float4 r[kLseRegQ];
    float m = -INFINITY;
#pragma unroll
    for (int k = 0; k < kLseRegQ; ++k) {
      const int q = k * 64 + lane;
      float4 v = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
      if (q < nq) {
        v = *reinterpret_cast<const float4*>(p + 4 * q);
        const int c = 4 * q;
        if (c + 1 >= V) v.y = -INFINITY;
        if (c + 2 >= V) v.z = -INFINITY;
        if (c + 3 >= V) v.w = -INFINITY;
      }
      r[k] = v;
      m = fmaxf(m, fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w)));
    }
    m = wave_max(m);
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < kLseRegQ; ++k) s += (expf(r[k].x - m) + expf(r[k].y - m)) + (expf(r[k].z - m) + expf(r[k].w - m));
    s = wave_add(s);
    if (lane == 0) lse[row] = m + logf(s);
    return;
  }
  float m = -INFINITY;
  for (int v = lane; v < V; v += 64) m = fmaxf(m, p[v]);
  m = wave_max(m);
  float s = 0.f;
  for (int v = lane; v < V; v += 64) s += expf(p[v] - m);
  s = wave_add(s);
  if (lane == 0) lse[row] = m + logf(s);
}

// one wave per utterance; extended label sequence l' (blank, y1, blank, y2, ..., blank), state s on lane s % 64,
// up to NC * 64 states.
// (round 6: the bodies are templates on the chunk count NC - 4 for targets up to 127 labels, kMaxChunks = 7 for up to 223, past the
// yaml's token_max_length of 200; labels of more than 127 tokens were refused until then)
constexpr int kMaxChunks = 7;
template <int NC>
__device__ __forceinline__ void ctc_alpha_body(const float* __restrict__ logits, int64_t ld, int T,
                                               const float* __restrict__ lse, const int32_t* __restrict__ ys,
                                               int Lmax, const int32_t* __restrict__ hlens,
                                               const int32_t* __restrict__ ylens, int blank,
                                               float* __restrict__ loss, float* __restrict__ alpha_out,
                                               int Smax) {
  const int b = blockIdx.x, lane = threadIdx.x;
  int tlen = hlens[b];
  if (tlen > T) tlen = T;
  const int U = ylens[b];
  const int S = 2 * U + 1;
  const int nch = (S + 63) / 64;
  int lab[NC];
  bool skip[NC];  // transition from s-2 allowed
  float alpha[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const int s = c * 64 + lane;
    int l = blank;
    bool sk = false;
    if (s < S && (s & 1)) {
      l = ys[(int64_t)b * Lmax + (s >> 1)];
      sk = (s >= 3) && (ys[(int64_t)b * Lmax + (s >> 1) - 1] != l);
    }
    lab[c] = l;
    skip[c] = sk;
    alpha[c] = -INFINITY;
  }
  if (tlen < 1 || U < 0 || nch > NC) {
    if (lane == 0) loss[b] = (tlen < 1 && U == 0) ? 0.0f : INFINITY;
    return;
  }
  const float* row0 = logits + (int64_t)b * T * ld;
  // t = 0: alpha(0) = logp(blank), alpha(1) = logp(y1)
  {
    const float z = lse[(int64_t)b * T];
    if (lane == 0) alpha[0] = row0[blank] - z;
    if (lane == 1 && S > 1) alpha[0] = row0[lab[0]] - z;
    if (alpha_out) {
#pragma unroll
      for (int c = 0; c < NC; ++c)
        if (c * 64 + lane < S) alpha_out[((int64_t)b * T) * Smax + c * 64 + lane] = alpha[c];
    }
  }
  // The emissions of a step do not depend on the recursion: they are fetched kPD steps ahead (as loads inside the step they were an
  // exposed L2 / HBM round trip per time step: 265 us for 255 steps of the cfg-4 batch, one wave per utterance).
  constexpr int kPD = 4;
  float er[kPD][NC], zr[kPD];
  auto fetch = [&](auto kc, int t) __attribute__((always_inline)) {
    constexpr int k = decltype(kc)::value;
    const int tc = t < tlen ? t : tlen - 1;
    const float* row = row0 + (int64_t)tc * ld;
    zr[k] = lse[(int64_t)b * T + tc];
#pragma unroll
    for (int c = 0; c < NC; ++c) er[k][c] = (c < nch && c * 64 + lane < S) ? row[lab[c]] : 0.0f;
  };
  auto step = [&](int t, const float (&e)[NC], float z) __attribute__((always_inline)) {
    float carry1 = -INFINITY, carry2 = -INFINITY;  // alpha(s-1), alpha(s-2) coming from the previous chunk
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      if (c >= nch) break;
      const float a0 = alpha[c];
      float a1 = __shfl_up(a0, 1, 64);
      float a2 = __shfl_up(a0, 2, 64);
      const float last1 = __shfl(a0, 63, 64), last2 = __shfl(a0, 62, 64);
      if (lane == 0) { a1 = carry1; a2 = carry2; }
      if (lane == 1) a2 = carry1;
      carry1 = last1;
      carry2 = last2;
      const float v = log_add3(a0, a1, a2, skip[c]);
      const int s = c * 64 + lane;
      alpha[c] = (s < S) ? v + (e[c] - z) : -INFINITY;
      if (alpha_out && s < S) alpha_out[((int64_t)b * T + t) * Smax + s] = alpha[c];
    }
  };
  using K0 = std::integral_constant<int, 0>;
  using K1 = std::integral_constant<int, 1>;
  using K2 = std::integral_constant<int, 2>;
  using K3 = std::integral_constant<int, 3>;
  fetch(K0{}, 1); fetch(K1{}, 2); fetch(K2{}, 3); fetch(K3{}, 4);
  int t = 1;
  for (; t + kPD <= tlen; t += kPD) {
    step(t, er[0], zr[0]); fetch(K0{}, t + kPD);
    step(t + 1, er[1], zr[1]); fetch(K1{}, t + 1 + kPD);
    step(t + 2, er[2], zr[2]); fetch(K2{}, t + 2 + kPD);
    step(t + 3, er[3], zr[3]); fetch(K3{}, t + 3 + kPD);
  }
  // (the ring holds steps t .. t + 3 in slots 0 .. 3 here: at most three of them are left)
  if (t < tlen) step(t, er[0], zr[0]);
  if (t + 1 < tlen) step(t + 1, er[1], zr[1]);
  if (t + 2 < tlen) step(t + 2, er[2], zr[2]);
  // -log( alpha_T(S-1) + alpha_T(S-2) )
  float fin = -INFINITY;
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const int s = c * 64 + lane;
    if (s == S - 1 || (s == S - 2 && S > 1)) fin = log_add(fin, alpha[c]);
  }
  float m = wave_max(fin);
  float sum = wave_add(fin == -INFINITY ? 0.0f : expf(fin - m));
  if (lane == 0) loss[b] = (m == -INFINITY) ? INFINITY : -(m + logf(sum));
}

template <int NC>
__global__ __launch_bounds__(64) void ctc_alpha_kernel(const float* __restrict__ logits, int64_t ld, int T,
                                                       const float* __restrict__ lse, const int32_t* __restrict__ ys,
                                                       int Lmax, const int32_t* __restrict__ hlens,
                                                       const int32_t* __restrict__ ylens, int blank,
                                                       float* __restrict__ loss, float* __restrict__ alpha_out,
                                                       int Smax) {
  ctc_alpha_body<NC>(logits, ld, T, lse, ys, Lmax, hlens, ylens, blank, loss, alpha_out, Smax);
}

// Backward, step 1: beta recursion of one utterance per wave (mirror image of the alpha recursion).  Round 4: it no longer reads
// alpha or the loss - it stores log beta_t(s) WITHOUT the step's emission, beta_out[b, t, s], and ctc_dlogits_kernel forms the state
// occupancy w_t(s) = alpha_t(s) beta_t(s) / (y_t(l'_s) P(l|x)) = exp(alpha + beta_out + nll) itself - so that the two recursions,
// 255 dependent steps of one wave per utterance each, run SIDE BY SIDE in one launch (ctc_alpha_beta_kernel) instead of one after
// the other (79 + 112 us of the training step with 40 of 1024 SIMDs busy).
template <int NC>
__device__ __forceinline__ void ctc_beta_body(const float* __restrict__ logits, int64_t ld, int T,
                                              const float* __restrict__ lse, const int32_t* __restrict__ ys,
                                              int Lmax, const int32_t* __restrict__ hlens,
                                              const int32_t* __restrict__ ylens, int blank, float* __restrict__ beta_out, int Smax) {
  const int b = blockIdx.x, lane = threadIdx.x;
  int tlen = hlens[b];
  if (tlen > T) tlen = T;
  const int U = ylens[b];
  const int S = 2 * U + 1;
  const int nch = (S + 63) / 64;
  if (tlen < 1 || U < 0 || nch > NC) return;  // no gradient (degenerate)
  int lab[NC];
  bool skip[NC];  // transition s -> s+2 allowed
  float beta[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const int s = c * 64 + lane;
    int l = blank;
    bool sk = false;
    if (s < S && (s & 1)) {
      l = ys[(int64_t)b * Lmax + (s >> 1)];
      sk = (s + 2 < S) && (ys[(int64_t)b * Lmax + (s >> 1) + 1] != l);
    }
    lab[c] = l;
    skip[c] = sk;
    beta[c] = -INFINITY;
  }
  const float* row0 = logits + (int64_t)b * T * ld;
  // as in the alpha recursion: emissions and log-sum-exp of a step are fetched kPD steps ahead of the recursion
  constexpr int kPD = 4;
  float er[kPD][NC], zr[kPD];
  auto fetch = [&](auto kc, int t) __attribute__((always_inline)) {
    constexpr int k = decltype(kc)::value;
    const int tc = t >= 0 ? t : 0;
    const float* row = row0 + (int64_t)tc * ld;
    zr[k] = lse[(int64_t)b * T + tc];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const bool in = c < nch && c * 64 + lane < S;
      er[k][c] = in ? row[lab[c]] : 0.0f;
    }
  };
  auto step = [&](int t, const float (&e)[NC], float z) __attribute__((always_inline)) {
    float carry1 = -INFINITY, carry2 = -INFINITY;  // beta_{t+1}(s+1), (s+2) coming from the next chunk
    float nb[NC];
#pragma unroll
    for (int c = NC - 1; c >= 0; --c) {
      nb[c] = -INFINITY;
      if (c >= nch) continue;
      const int s = c * 64 + lane;
      const float lp = e[c] - z;
      float v;
      if (t == tlen - 1) {
        v = (s == S - 1 || (s == S - 2 && S > 1)) ? 0.0f : -INFINITY;
      } else {
        const float b0 = beta[c];
        float b1 = __shfl_down(b0, 1, 64);
        float b2 = __shfl_down(b0, 2, 64);
        const float first1 = __shfl(b0, 0, 64), first2 = __shfl(b0, 1, 64);
        if (lane == 63) { b1 = carry1; b2 = carry2; }
        if (lane == 62) b2 = carry1;
        carry1 = first1;
        carry2 = first2;
        v = log_add3(b0, b1, b2, skip[c]);
      }
      nb[c] = (s < S) ? v + lp : -INFINITY;
      if (s < S) beta_out[((int64_t)b * T + t) * Smax + s] = nb[c] - lp;  // (v: -inf stays -inf)
    }
#pragma unroll
    for (int c = 0; c < NC; ++c) beta[c] = nb[c];
  };
  using K0 = std::integral_constant<int, 0>;
  using K1 = std::integral_constant<int, 1>;
  using K2 = std::integral_constant<int, 2>;
  using K3 = std::integral_constant<int, 3>;
  int t = tlen - 1;
  fetch(K0{}, t); fetch(K1{}, t - 1); fetch(K2{}, t - 2); fetch(K3{}, t - 3);
  for (; t - kPD + 1 >= 0; t -= kPD) {
    step(t, er[0], zr[0]); fetch(K0{}, t - kPD);
    step(t - 1, er[1], zr[1]); fetch(K1{}, t - 1 - kPD);
    step(t - 2, er[2], zr[2]); fetch(K2{}, t - 2 - kPD);
    step(t - 3, er[3], zr[3]); fetch(K3{}, t - 3 - kPD);
  }
  if (t >= 0) step(t, er[0], zr[0]);
  if (t - 1 >= 0) step(t - 1, er[1], zr[1]);
  if (t - 2 >= 0) step(t - 2, er[2], zr[2]);
}

// both recursions in one launch: grid (batch, 2), blockIdx.y = 0: alpha (+ the utterance's loss), 1: beta
template <int NC>
__global__ __launch_bounds__(64) void ctc_alpha_beta_kernel(const float* __restrict__ logits, int64_t ld, int T,
                                                            const float* __restrict__ lse, const int32_t* __restrict__ ys,
                                                            int Lmax, const int32_t* __restrict__ hlens,
                                                            const int32_t* __restrict__ ylens, int blank, float* __restrict__ loss,
                                                            float* __restrict__ alpha_out, float* __restrict__ beta_out, int Smax) {
  if (blockIdx.y == 0) ctc_alpha_body<NC>(logits, ld, T, lse, ys, Lmax, hlens, ylens, blank, loss, alpha_out, Smax);
  else ctc_beta_body<NC>(logits, ld, T, lse, ys, Lmax, hlens, ylens, blank, beta_out, Smax);
}

// Backward, step 2: one workgroup per (b, t) row: dlogits[v] = scale * (softmax[v] - sum_{s: l'_s = v} w_t(s)) as bf16;
// rows past the utterance's length, and utterances with an infinite loss (zero_infinity), get zeros.  Columns
// [V, ld_out) are zeroed (GEMM K padding).
__device__ __forceinline__ uint32_t dlogit_bits(float g) {  // bf16, round to nearest even
  uint32_t u = __float_as_uint(g);
  u += 0x7fffu + ((u >> 16) & 1u);
  return u >> 16;
}
__device__ __forceinline__ void st_dlogit(uint16_t* p, float g) { *p = (uint16_t)dlogit_bits(g); }
__device__ __forceinline__ void st_dlogit(float* p, float g) { *p = g; }  // float32 validation mode (ma_ctc_loss_grad_x32)
template <typename OT>
__global__ __launch_bounds__(256) void ctc_dlogits_kernel(const float* __restrict__ logits, int64_t ld, int T, int V,
                                                          const float* __restrict__ lse, const int32_t* __restrict__ ys,
                                                          int Lmax, const int32_t* __restrict__ hlens,
                                                          const int32_t* __restrict__ ylens, int blank,
                                                          const float* __restrict__ loss, const float* __restrict__ ab,
                                                          const float* __restrict__ bb, int Smax, float scale,
                                                          OT* __restrict__ out, int64_t ld_out) {
  extern __shared__ float occ[];
  const int64_t row = blockIdx.x;
  const int b = (int)(row / T), t = (int)(row - (int64_t)b * T);
  OT* o = out + row * ld_out;
  int tlen = hlens[b];
  if (tlen > T) tlen = T;
  const int U = ylens[b], S = 2 * U + 1;
  if (t >= tlen || isinf(loss[b]) || S > Smax) {
    for (int v = threadIdx.x; v < ld_out; v += 256) o[v] = 0;
    return;
  }
  for (int v = threadIdx.x; v < V; v += 256) occ[v] = 0.0f;
  __syncthreads();
  // occ[v] = sum of the occupancies of the states that emit v, in a FIXED order (run-to-run deterministic; float atomics on the
  // LDS word of a repeated label / of the blank gave the sum whatever order the hardware served them in).
  //   wave 0: the blank (every even state, and any label equal to the blank): lane-strided sums + a shuffle tree;
  //   threads 64 ..: label position i: the FIRST occurrence of a label adds all its occurrences in index order.
  const float* abr = ab + row * Smax;
  const float* bbr = bb + row * Smax;
  const float nll = loss[b];
  // w_t(s) = alpha_t(s) beta_t(s) / (y_t(l'_s) P(l|x)): log alpha and log beta (without the emission) from the two recursions
  auto occ_at = [&](int s) -> float {
    const float ev = abr[s] + bbr[s] + nll;
    return ev > -80.0f ? expf(ev) : 0.0f;
  };
  const int32_t* yb = ys + (int64_t)b * Lmax;
  if (threadIdx.x < 64) {
    float a = 0.0f;
    for (int s = threadIdx.x; s < S; s += 64)
      if (!(s & 1) || yb[s >> 1] == blank) a += occ_at(s);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) a += __shfl_xor(a, off, 64);
    if (threadIdx.x == 0) occ[blank] = a;
  } else {
    for (int i = threadIdx.x - 64; i < U; i += 192) {
      const int l = yb[i];
      if (l == blank) continue;
      bool first = true;
      for (int k = 0; k < i; ++k) first = first && (yb[k] != l);
      if (!first) continue;
      float a = 0.0f;
      for (int k = i; k < U; ++k)
        if (yb[k] == l) a += occ_at(2 * k + 1);
      occ[l] = a;
    }
  }
  __syncthreads();
  const float* p = logits + row * ld;
  const float z = lse[row];
  if constexpr (std::is_same<OT, uint16_t>::value) {
    // four columns per thread: one 16-byte load, one 8-byte store (round 4; 4-byte loads and 2-byte stores ran at 3.4 TB/s)
    if ((ld & 3) == 0 && (ld_out & 3) == 0 && ((reinterpret_cast<uintptr_t>(logits) | reinterpret_cast<uintptr_t>(out)) & 15) == 0 &&
        ld >= ((V + 3) & ~3)) {
      for (int v = threadIdx.x * 4; v < ld_out; v += 1024) {
        float g4[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        if (v < V) {
          const float4 x = *reinterpret_cast<const float4*>(p + v);
          const float xs[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (v + j < V) g4[j] = scale * (expf(xs[j] - z) - occ[v + j]);
        }
        uint2 w;
        w.x = dlogit_bits(g4[0]) | (dlogit_bits(g4[1]) << 16);
        w.y = dlogit_bits(g4[2]) | (dlogit_bits(g4[3]) << 16);
        *reinterpret_cast<uint2*>(o + v) = w;
      }
      return;
    }
  }
  for (int v = threadIdx.x; v < ld_out; v += 256) {
    float gval = 0.0f;
    if (v < V) gval = scale * (expf(p[v] - z) - occ[v]);
    st_dlogit(o + v, gval);
  }
}

// CTC greedy search (mindaudio/models/decoders/decoder_factory.py:9-56): one wave per frame row: argmax over the
// vocabulary (first index on ties, like TopK) of the logits = argmax of the log-softmax, its log-probability, and the
// frame index masked by the encoder mask (padded frames -> 0 = blank).
__global__ __launch_bounds__(256) void ctc_greedy_kernel(const float* __restrict__ logits, int64_t ld, int64_t rows, int V,
                                                         const float* __restrict__ mask, int32_t* __restrict__ best,
                                                         float* __restrict__ best_logp) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* p = logits + row * ld;
  float m = -INFINITY;
  int am = 0;
  for (int v = lane; v < V; v += 64)
    if (p[v] > m) { m = p[v]; am = v; }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const float om = __shfl_xor(m, off, 64);
    const int oa = __shfl_xor(am, off, 64);
    if (om > m || (om == m && oa < am)) { m = om; am = oa; }
  }
  float s = 0.f;
  for (int v = lane; v < V; v += 64) s += expf(p[v] - m);
  s = wave_add(s);
  if (lane == 0) {
    const bool keep = !mask || mask[row] != 0.0f;
    best[row] = keep ? am : 0;
    best_logp[row] = -logf(s);  // logp(argmax) = max - (max + log sum exp(p - max))
  }
}

// remove_duplicates_and_blank (mindaudio/utils/common.py:116-125): collapse repeats, drop blanks; one thread per
// utterance (T is a few hundred frames).  hyp (batch, T) int32 zero-padded, hyp_len (batch).
__global__ void ctc_collapse_kernel(const int32_t* __restrict__ best, int B, int T, int blank, int32_t* __restrict__ hyp,
                                    int32_t* __restrict__ hyp_len) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const int32_t* src = best + (int64_t)b * T;
  int32_t* dst = hyp + (int64_t)b * T;
  int n = 0, prev = -1;
  for (int t = 0; t < T; ++t) {
    const int v = src[t];
    if (v != prev && v != blank) dst[n++] = v;
    prev = v;
  }
  hyp_len[b] = n;
  for (int t = n; t < T; ++t) dst[t] = 0;
}

__global__ void ctc_reduce_kernel(const float* __restrict__ loss, int B, int zero_infinity, float* __restrict__ out) {
  // fixed-order sum over the batch (deterministic), zero_infinity as CTCLossV2, then / B (ctc_loss.py:61-62)
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    float s = 0.f;
    for (int b = 0; b < B; ++b) {
      float v = loss[b];
      if (zero_infinity && isinf(v)) v = 0.0f;
      s += v;
    }
    out[0] = s / (float)B;
  }
}

__global__ __launch_bounds__(256) void cast_f32_bf16_kernel(const float* __restrict__ x, uint16_t* __restrict__ y, int64_t n4) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    const float4 v = reinterpret_cast<const float4*>(x)[i];
    uint32_t lo, hi;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(lo) : "v"(v.x), "v"(v.y));
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(hi) : "v"(v.z), "v"(v.w));
    reinterpret_cast<uint2*>(y)[i] = make_uint2(lo, hi);
  }
}

}  // namespace ma

using namespace ma;

extern "C" {

int ma_ctc_loss_f32(const float* logits, int64_t ld, int64_t batch, int64_t T, int32_t V, const int32_t* ys,
                    int32_t Lmax, const int32_t* hlens, const int32_t* ylens, int32_t blank, int32_t zero_infinity,
                    float* per_utt_loss, float* lse_workspace, float* loss_out, ma_stream_t stream) {
  if (!logits || !ys || !hlens || !ylens || !per_utt_loss || !lse_workspace || !loss_out) return MA_ERR_INVALID_ARG;
  if (batch < 1 || T < 1 || V < 1 || ld < V || Lmax < 1 || blank < 0 || blank >= V) return MA_ERR_INVALID_ARG;
  if (2 * Lmax + 1 > kMaxChunks * 64) return MA_ERR_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  const int64_t rows = batch * T;
  MA_LAUNCH(ctc_lse_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, logits, ld, rows, (int)V, lse_workspace);
  if (2 * Lmax + 1 <= 4 * 64)  // (the 4-chunk form: its per-lane state fits the registers of the common case, targets up to 127 labels)
    MA_LAUNCH(ctc_alpha_kernel<4>, dim3((unsigned)batch), dim3(64), 0, s, logits, ld, (int)T, lse_workspace, ys, (int)Lmax,
              hlens, ylens, (int)blank, per_utt_loss, (float*)nullptr, 0);
  else
    MA_LAUNCH(ctc_alpha_kernel<kMaxChunks>, dim3((unsigned)batch), dim3(64), 0, s, logits, ld, (int)T, lse_workspace, ys, (int)Lmax,
              hlens, ylens, (int)blank, per_utt_loss, (float*)nullptr, 0);
  MA_LAUNCH(ctc_reduce_kernel, dim3(1), dim3(64), 0, s, per_utt_loss, (int)batch, (int)zero_infinity, loss_out);
  return MA_OK;
}

int64_t ma_ctc_grad_workspace_bytes(int64_t batch, int64_t T, int32_t Lmax) {
  if (batch < 1 || T < 1 || Lmax < 1) return MA_ERR_INVALID_ARG;
  return 2 * batch * T * (2 * (int64_t)Lmax + 1) * 4;  // log alpha | log beta
}

extern "C++" {
template <typename OT>
static int ctc_loss_grad_launch(const float* logits, int64_t ld, int64_t batch, int64_t T, int32_t V, const int32_t* ys,
                                int32_t Lmax, const int32_t* hlens, const int32_t* ylens, int32_t blank,
                                int32_t zero_infinity, float grad_scale, float* per_utt_loss, float* lse_workspace,
                                float* loss_out, OT* dlogits, int64_t ld_out, void* workspace, int64_t workspace_bytes,
                                ma_stream_t stream) {
  if (!logits || !ys || !hlens || !ylens || !per_utt_loss || !lse_workspace || !loss_out || !dlogits || !workspace)
    return MA_ERR_INVALID_ARG;
  if (batch < 1 || T < 1 || V < 1 || ld < V || ld_out < V || Lmax < 1 || blank < 0 || blank >= V) return MA_ERR_INVALID_ARG;
  if (2 * Lmax + 1 > kMaxChunks * 64 || !zero_infinity || (int64_t)V * 4 > 60 * 1024) return MA_ERR_UNSUPPORTED;
  if (workspace_bytes < ma_ctc_grad_workspace_bytes(batch, T, Lmax)) return MA_ERR_WORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  const int64_t rows = batch * T;
  const int Smax = 2 * Lmax + 1;
  float* ab = reinterpret_cast<float*>(workspace);
  MA_LAUNCH(ctc_lse_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, logits, ld, rows, (int)V, lse_workspace);
  float* bb = ab + batch * T * Smax;
  if (Smax <= 4 * 64)
    MA_LAUNCH(ctc_alpha_beta_kernel<4>, dim3((unsigned)batch, 2), dim3(64), 0, s, logits, ld, (int)T, lse_workspace, ys, (int)Lmax,
              hlens, ylens, (int)blank, per_utt_loss, ab, bb, Smax);
  else
    MA_LAUNCH(ctc_alpha_beta_kernel<kMaxChunks>, dim3((unsigned)batch, 2), dim3(64), 0, s, logits, ld, (int)T, lse_workspace, ys,
              (int)Lmax, hlens, ylens, (int)blank, per_utt_loss, ab, bb, Smax);
  MA_LAUNCH(ctc_reduce_kernel, dim3(1), dim3(64), 0, s, per_utt_loss, (int)batch, 1, loss_out);
  MA_LAUNCH(ctc_dlogits_kernel<OT>, dim3((unsigned)rows), dim3(256), (size_t)V * 4, s, logits, ld, (int)T, (int)V,
            lse_workspace, ys, (int)Lmax, hlens, ylens, (int)blank, per_utt_loss, ab, bb, Smax, grad_scale, dlogits, ld_out);
  return MA_OK;
}
}  // extern "C++"
int ma_ctc_loss_grad_f32(const float* logits, int64_t ld, int64_t batch, int64_t T, int32_t V, const int32_t* ys,
                         int32_t Lmax, const int32_t* hlens, const int32_t* ylens, int32_t blank,
                         int32_t zero_infinity, float grad_scale, float* per_utt_loss, float* lse_workspace,
                         float* loss_out, void* dlogits, int64_t ld_out, void* workspace, int64_t workspace_bytes,
                         ma_stream_t stream) {
  return ctc_loss_grad_launch(logits, ld, batch, T, V, ys, Lmax, hlens, ylens, blank, zero_infinity, grad_scale, per_utt_loss,
                              lse_workspace, loss_out, reinterpret_cast<uint16_t*>(dlogits), ld_out, workspace, workspace_bytes,
                              stream);
}
int ma_ctc_loss_grad_x32(const float* logits, int64_t ld, int64_t batch, int64_t T, int32_t V, const int32_t* ys,
                         int32_t Lmax, const int32_t* hlens, const int32_t* ylens, int32_t blank,
                         int32_t zero_infinity, float grad_scale, float* per_utt_loss, float* lse_workspace,
                         float* loss_out, float* dlogits, int64_t ld_out, void* workspace, int64_t workspace_bytes,
                         ma_stream_t stream) {
  return ctc_loss_grad_launch(logits, ld, batch, T, V, ys, Lmax, hlens, ylens, blank, zero_infinity, grad_scale, per_utt_loss,
                              lse_workspace, loss_out, dlogits, ld_out, workspace, workspace_bytes, stream);
}

int ma_ctc_greedy_search_f32(const float* logits, int64_t ld, int64_t batch, int64_t T, int32_t V, const float* mask,
                             int32_t blank, int32_t* best, float* best_logp, int32_t* hyp, int32_t* hyp_len,
                             ma_stream_t stream) {
  if (!logits || !best || !best_logp || !hyp || !hyp_len || batch < 1 || T < 1 || V < 1 || ld < V || blank < 0 || blank >= V)
    return MA_ERR_INVALID_ARG;
  hipStream_t s = (hipStream_t)stream;
  const int64_t rows = batch * T;
  MA_LAUNCH(ctc_greedy_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, logits, ld, rows, (int)V, mask, best, best_logp);
  MA_LAUNCH(ctc_collapse_kernel, dim3((unsigned)((batch + 63) / 64)), dim3(64), 0, s, best, (int)batch, (int)T, (int)blank, hyp,
            hyp_len);
  return MA_OK;
}

int ma_cast_f32_bf16(const float* x, void* y, int64_t n, ma_stream_t stream) {
  if (!x || !y || n < 1 || (n & 3)) return MA_ERR_INVALID_ARG;
  int64_t blocks = (n / 4 + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  MA_LAUNCH(cast_f32_bf16_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x,
            reinterpret_cast<uint16_t*>(y), n / 4);
  return MA_OK;
}

}  // extern "C"
