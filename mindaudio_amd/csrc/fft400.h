// 400-point real FFT on 8 lanes of a wavefront (gfx950, wave64): a wave transforms EIGHT frames at once - the mixed-radix sibling
// of fft512.h for the reference's DEFAULT n_fft (mindaudio/data/features.py:201, spectrum.py:611: n_fft = 400).
//
// Per frame (lane l = 0..7 of its 8-lane group), z[m] = x[2m] + i x[2m+1], m = 8 m1 + m2, 200 = 25 x 8:
//   pass 1  lane l owns column m2 = l: ONE 25-point FFT over m1 in registers (5 x 5 radix-5 butterflies), then the twiddle
//           W200^(q l).
//   swap    the 25 x 8 complex matrix goes through the frame's LDS slot, one component at a time (4-byte writes of a column,
//           16-byte reads of whole rows) - the only cross-lane exchange.
//   pass 2  a lane owns up to two row PAIRS (q, 25 - q): 8-point FFTs over m2 give Z[q + 25 r] and Z[(25 - q) + 25 r], r = 0..7,
//           so Z[k] and Z[200 - k] sit in the SAME lane and the real-FFT split (fft512.h: rsplit_pair) needs no lane traffic:
//               lanes 0-3: pairs (l + 1, 24 - l) and (l + 5, 20 - l);   lanes 4-6: pair (l + 5, 20 - l), computed twice;
//               lane 7: pair (12, 13) and row 0, which pairs with itself (r <-> 8 - r; 16 selects).
//           Bins 0..200: 12 pairs x 8 + the row-0 bins 0, 25, .., 200.
// All arithmetic is planar scalar f32; the window is pre-scaled by 1/2 by the caller, which makes the split outputs exactly X.
#pragma once
#include <hip/hip_runtime.h>

#include "fft512.h"

namespace ma {

// 5-point DFT (forward, e^{-2 pi i n k / 5}), natural order in and out
__device__ __forceinline__ void radix5(float& r0, float& i0, float& r1, float& i1, float& r2, float& i2, float& r3, float& i3,
                                       float& r4, float& i4) {
  constexpr float CA = 0.30901699437494742f;   // cos(2 pi / 5)
  constexpr float CB = -0.80901699437494742f;  // cos(4 pi / 5)
  constexpr float SA = 0.95105651629515357f;   // sin(2 pi / 5)
  constexpr float SB = 0.58778525229247313f;   // sin(4 pi / 5)
  const float t1r = r1 + r4, t1i = i1 + i4, t3r = r1 - r4, t3i = i1 - i4;
  const float t2r = r2 + r3, t2i = i2 + i3, t4r = r2 - r3, t4i = i2 - i3;
  const float p1r = r0 + CA * t1r + CB * t2r, p1i = i0 + CA * t1i + CB * t2i;
  const float p2r = r0 + CB * t1r + CA * t2r, p2i = i0 + CB * t1i + CA * t2i;
  const float q1r = SA * t3r + SB * t4r, q1i = SA * t3i + SB * t4i;
  const float q2r = SB * t3r - SA * t4r, q2i = SB * t3i - SA * t4i;
  r0 = r0 + t1r + t2r; i0 = i0 + t1i + t2i;
  r1 = p1r + q1i; i1 = p1i - q1r;  // p1 - i q1
  r4 = p1r - q1i; i4 = p1i + q1r;  // p1 + i q1
  r2 = p2r + q2i; i2 = p2i - q2r;  // p2 - i q2
  r3 = p2r - q2i; i3 = p2i + q2r;  // p2 + i q2
}

// where X[k] of fft25 lives
__host__ __device__ constexpr int pos25(int k) { return 5 * (k % 5) + k / 5; }

// forward 25-point DFT, input natural order (index n = 5 n1 + n2), output X[k] at index pos25(k)
__device__ __forceinline__ void fft25(float (&r)[25], float (&i)[25]) {
  // step A: 5-point DFTs over n1 for each n2: elements n2, 5 + n2, .., 20 + n2 -> A[k1][n2] at index 5 k1 + n2
#pragma unroll
  for (int n2 = 0; n2 < 5; ++n2)
    radix5(r[n2], i[n2], r[5 + n2], i[5 + n2], r[10 + n2], i[10 + n2], r[15 + n2], i[15 + n2], r[20 + n2], i[20 + n2]);
  // step B: A[k1][n2] *= W25^(n2 k1)
  // (cos, sin)(2 pi e / 25) at the exponents e = k1 n2 that occur: 1, 2, 3, 4, 6, 8, 9, 12, 16
  constexpr float C[17] = {1.0f, 0.96858316112863108f, 0.87630668004386358f, 0.72896862742141155f, 0.53582679497899655f, 0.0f, 0.062790519529313527f, 0.0f, -0.42577929156507272f, -0.63742398974868975f, 0.0f, 0.0f, -0.99211470131447776f, 0.0f, 0.0f, 0.0f, -0.63742398974868952f};
  constexpr float S[17] = {0.0f, 0.24868988716485479f, 0.48175367410171532f, 0.68454710592868862f, 0.84432792550201508f, 0.0f, 0.99802672842827156f, 0.0f, 0.90482705246601947f, 0.77051324277578925f, 0.0f, 0.0f, 0.12533323356430454f, 0.0f, 0.0f, 0.0f, -0.77051324277578936f};
#pragma unroll
  for (int k1 = 1; k1 < 5; ++k1)
#pragma unroll
    for (int n2 = 1; n2 < 5; ++n2) {
      const int e = k1 * n2;  // 1, 2, 3, 4, 6, 8, 9, 12, 16
      cmul_inplace(r[5 * k1 + n2], i[5 * k1 + n2], C[e], -S[e]);
    }
  // step C: 5-point DFTs over n2 for each k1 -> X[k1 + 5 k2] at index 5 k1 + k2
#pragma unroll
  for (int k1 = 0; k1 < 5; ++k1)
    radix5(r[5 * k1], i[5 * k1], r[5 * k1 + 1], i[5 * k1 + 1], r[5 * k1 + 2], i[5 * k1 + 2], r[5 * k1 + 3], i[5 * k1 + 3],
           r[5 * k1 + 4], i[5 * k1 + 4]);
}

// forward 8-point DFT in place, natural order in, X[k] at index rev3(k) (bit reversal)
__host__ __device__ constexpr int rev3(int k) { return ((k & 1) << 2) | (k & 2) | ((k >> 2) & 1); }
__device__ __forceinline__ void fft8(float (&r)[8], float (&i)[8]) {
  constexpr float H = 0.70710678118654752f;
  // stage 1: span 4, twiddles W8^j on the differences
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float ar = r[j] + r[j + 4], ai = i[j] + i[j + 4], br = r[j] - r[j + 4], bi = i[j] - i[j + 4];
    r[j] = ar; i[j] = ai; r[j + 4] = br; i[j + 4] = bi;
  }
  { const float a = r[5], b = i[5]; r[5] = (a + b) * H; i[5] = (b - a) * H; }    // W8^1 = (1 - i) / sqrt 2
  { const float a = r[6]; r[6] = i[6]; i[6] = -a; }                              // W8^2 = -i
  { const float a = r[7], b = i[7]; r[7] = (b - a) * H; i[7] = -(a + b) * H; }   // W8^3 = (-1 - i) / sqrt 2
  // stages 2 + 3: a 4-point DFT on each half: outputs X[0], X[2], X[1], X[3] of the half at its indices 0, 1, 2, 3
#pragma unroll
  for (int h = 0; h < 8; h += 4) {
    const float s0r = r[h] + r[h + 2], s0i = i[h] + i[h + 2], s1r = r[h] - r[h + 2], s1i = i[h] - i[h + 2];
    const float s2r = r[h + 1] + r[h + 3], s2i = i[h + 1] + i[h + 3], s3r = r[h + 1] - r[h + 3], s3i = i[h + 1] - i[h + 3];
    r[h] = s0r + s2r; i[h] = s0i + s2i;          // half-DFT bin 0
    r[h + 1] = s0r - s2r; i[h + 1] = s0i - s2i;  // bin 2
    r[h + 2] = s1r + s3i; i[h + 2] = s1i - s3r;  // bin 1: s1 - i s3
    r[h + 3] = s1r - s3i; i[h + 3] = s1i + s3r;  // bin 3: s1 + i s3
  }
  // index h + {0, 1, 2, 3} holds half-bins {0, 2, 1, 3}; the first half is X[2 b], the second X[2 b + 1]:
  //   index 0..7 = X[0], X[4], X[2], X[6], X[1], X[5], X[3], X[7]  = X[rev3(index)]
}

constexpr int kBins400 = 201;

struct Rfft400Lane {
  int l;
  int row[4];   // A0, B0, A1, B1: rows of the 25 x 8 matrix this lane transforms in pass 2 (floats offset = row * 8)
  bool lane7;   // second pair is row 0 with itself
};

__device__ __forceinline__ Rfft400Lane rfft400_lane_setup(int lane) {
  Rfft400Lane s;
  const int l = lane & 7;
  s.l = l;
  s.lane7 = (l == 7);
  const int a0 = l < 4 ? l + 1 : l + 5;
  const int a1 = l < 4 ? l + 5 : (l == 7 ? 0 : a0);
  s.row[0] = a0;
  s.row[1] = 25 - a0;
  s.row[2] = a1;
  s.row[3] = l < 4 ? 25 - a1 : (l == 7 ? 0 : 25 - a0);
  return s;
}

// 400-point real FFT of 8 frames per wave.
//   In : (zr, zi)[m1] = windowed (x[16 m1 + 2l], x[16 m1 + 2l + 1]) * 1/2
//   Out: emit(pair, r, xr, xi, yr, yi) for pair = 0, 1 and r = 0..7 (compile-time constants): (xr, xi) = X[k], (yr, yi) = X[200 - k],
//        k = row[2 pair] + 25 r.
//   tw200: LDS float2 table [q * 8 + l] = W200^(q l) (cos, -sin);  tw400: LDS float2 table [k] = (cos, sin)(2 pi k / 400), k = 0..200
//   slot : this frame's LDS area (>= 200 floats, 16-byte aligned)
template <class Emit>
__device__ __forceinline__ void rfft400_x8(float (&zr)[25], float (&zi)[25], const Rfft400Lane& s,
                                           const float2* __restrict__ tw200, const float2* __restrict__ tw400,
                                           float* __restrict__ slot, Emit&& emit) {
  __builtin_amdgcn_sched_barrier(0);
  fft25(zr, zi);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int q = 1; q < 25; ++q) {
    const float2 w = tw200[q * 8 + s.l];
    cmul_inplace(zr[pos25(q)], zi[pos25(q)], w.x, w.y);
  }
  __builtin_amdgcn_sched_barrier(0);
  // ---- 25 x 8 exchange through LDS, real parts then imaginary parts ----------------------------------------------------
  float ar[4][8], ai[4][8];  // [A0, B0, A1, B1][m2]
#pragma unroll
  for (int q = 0; q < 25; ++q) slot[q * 8 + s.l] = zr[pos25(q)];
  wave_lds_sync();
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const float4 lo = *reinterpret_cast<const float4*>(slot + s.row[c] * 8);
    const float4 hi = *reinterpret_cast<const float4*>(slot + s.row[c] * 8 + 4);
    ar[c][0] = lo.x; ar[c][1] = lo.y; ar[c][2] = lo.z; ar[c][3] = lo.w;
    ar[c][4] = hi.x; ar[c][5] = hi.y; ar[c][6] = hi.z; ar[c][7] = hi.w;
  }
  wave_lds_sync();
#pragma unroll
  for (int q = 0; q < 25; ++q) slot[q * 8 + s.l] = zi[pos25(q)];
  wave_lds_sync();
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const float4 lo = *reinterpret_cast<const float4*>(slot + s.row[c] * 8);
    const float4 hi = *reinterpret_cast<const float4*>(slot + s.row[c] * 8 + 4);
    ai[c][0] = lo.x; ai[c][1] = lo.y; ai[c][2] = lo.z; ai[c][3] = lo.w;
    ai[c][4] = hi.x; ai[c][5] = hi.y; ai[c][6] = hi.z; ai[c][7] = hi.w;
  }
  wave_lds_sync();
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int c = 0; c < 4; ++c) fft8(ar[c], ai[c]);  // [c][rev3(r)] = Z[row[c] + 25 r]
  __builtin_amdgcn_sched_barrier(0);

  // ---- real-FFT split: A[r] with B[7 - r]; lane 7's second pair is row 0 with itself: A1[r] with A1[(8 - r) & 7] --------------
#pragma unroll
  for (int pair = 0; pair < 2; ++pair) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const float ur = ar[2 * pair][rev3(r)], ui = ai[2 * pair][rev3(r)];
      float vr = ar[2 * pair + 1][rev3(7 - r)], vi = ai[2 * pair + 1][rev3(7 - r)];
      if (pair == 1) {
        vr = s.lane7 ? ar[2][rev3((8 - r) & 7)] : vr;
        vi = s.lane7 ? ai[2][rev3((8 - r) & 7)] : vi;
      }
      const float2 w = tw400[s.row[2 * pair] + 25 * r];
      float xr, xi, yr, yi;
      rsplit_pair(ur, ui, vr, vi, w.x, w.y, xr, xi, yr, yi);
      emit(pair, r, xr, xi, yr, yi);
    }
  }
}

}  // namespace ma
