// 512-point real FFT on 8 lanes of a wavefront (gfx950, wave64): a wave transforms EIGHT frames at once.
//
// Per frame (lane l = 0..7 of its 8-lane group), with z[m] = x[2m] + i x[2m+1], m = 16 m1 + m2:
//   pass 1  lane l owns the two columns m2 = 2l, 2l+1 (one 16-byte load per m1 brings both): two 16-point
//           FFTs over m1 in registers, then the twiddle W256^(q m2).
//   swap    the 16x16 complex matrix goes through a 256-float LDS slot, one component at a time (re, im);
//           8-byte writes, 16-byte reads, chunks XOR-swizzled per frame (bank-conflict-free, see kSlotStride).  This is the
//           ONLY cross-lane exchange.
//   pass 2  lane l owns the column PAIR (k1, 16-k1) (lane 0: columns 8 and 0): two 16-point FFTs over m2.
//           Z[k] and Z[256-k] now sit in the SAME lane, so the real-FFT split
//               A = Z[k] + conj(Z[256-k]),  B = Z[k] - conj(Z[256-k]),  T = i W512^k B,
//               X[k] = A - T,   X[256-k] = conj(A + T)
//           needs no lane traffic and each twiddle serves two outputs.  (The window is pre-scaled by 1/2 by
//           the caller, which makes these exactly X, not 2X.)
//   Lane 0's columns pair with themselves (0: r <-> 16-r, 8: r <-> 15-r); 32 selects put its operands in
//   the generic slots, and X[128] is a one-liner.
// All arithmetic is planar scalar f32 (packed f32 ops buy no throughput on gfx950 and cost register moves).
#pragma once
#include <hip/hip_runtime.h>

namespace ma {

// base-4 digit reversal of a 4-bit index: where X[k] lives after fft16
__host__ __device__ constexpr int rev4(int p) { return ((p >> 2) & 3) | ((p & 3) << 2); }

__device__ __forceinline__ void radix4(float& r0, float& i0, float& r1, float& i1, float& r2, float& i2,
                                       float& r3, float& i3) {
  const float s0r = r0 + r2, s0i = i0 + i2, s1r = r0 - r2, s1i = i0 - i2;
  const float s2r = r1 + r3, s2i = i1 + i3, s3r = r1 - r3, s3i = i1 - i3;
  r0 = s0r + s2r; i0 = s0i + s2i;
  r2 = s0r - s2r; i2 = s0i - s2i;
  r1 = s1r + s3i; i1 = s1i - s3r;  // s1 - i s3
  r3 = s1r - s3i; i3 = s1i + s3r;  // s1 + i s3
}

__device__ __forceinline__ void cmul_inplace(float& r, float& i, float wr, float wi) {
  const float nr = r * wr - i * wi;
  i = r * wi + i * wr;
  r = nr;
}

// forward 16-point DFT, input natural order, output X[k] at index rev4(k)
__device__ __forceinline__ void fft16(float (&r)[16], float (&i)[16]) {
  constexpr float C1 = 0.92387953251128674f;  // cos(pi/8)
  constexpr float S1 = 0.38268343236508977f;  // sin(pi/8)
  constexpr float H = 0.70710678118654752f;
#pragma unroll
  for (int n2 = 0; n2 < 4; ++n2)
    radix4(r[n2], i[n2], r[4 + n2], i[4 + n2], r[8 + n2], i[8 + n2], r[12 + n2], i[12 + n2]);
  // element [4*k1 + n2] *= W16^(n2*k1)
  cmul_inplace(r[5], i[5], C1, -S1);
  { const float a = r[6], b = i[6]; r[6] = (a + b) * H; i[6] = (b - a) * H; }
  cmul_inplace(r[7], i[7], S1, -C1);
  { const float a = r[9], b = i[9]; r[9] = (a + b) * H; i[9] = (b - a) * H; }
  { const float a = r[10]; r[10] = i[10]; i[10] = -a; }
  { const float a = r[11], b = i[11]; r[11] = (b - a) * H; i[11] = -(a + b) * H; }
  cmul_inplace(r[13], i[13], S1, -C1);
  { const float a = r[14], b = i[14]; r[14] = (b - a) * H; i[14] = -(a + b) * H; }
  cmul_inplace(r[15], i[15], -C1, S1);
#pragma unroll
  for (int k1 = 0; k1 < 4; ++k1)
    radix4(r[4 * k1], i[4 * k1], r[4 * k1 + 1], i[4 * k1 + 1], r[4 * k1 + 2], i[4 * k1 + 2], r[4 * k1 + 3],
           i[4 * k1 + 3]);
}

constexpr int kSlotFloats = 256;  // one component of the 16x16 matrix

__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// LDS layout of the 16x16 exchange (one float component at a time).  Frame f's slot starts at f * kSlotStride floats and holds
// row q (the pass-1 output index) as 16 floats = four 16-byte chunks; chunk c of every row of frame f sits at chunk position
// c ^ (f & 3).  With the banking of gfx950 (MI355X_MICROARCH.md, LDS table):
//   * pass-1 stores are ds_write_b64, serviced in groups of 16 consecutive lanes = two frames x one row = two 64-byte runs, 32
//     banks: kSlotStride = 16 (mod 32) puts the two runs on disjoint banks;
//   * pass-2 loads are ds_read_b128 of whole rows, serviced in the lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31}, ... = four
//     lanes of each of four frames: the four lanes of a frame read four rows with distinct (row & 3) (different 64-byte bank
//     blocks), and inside a block the four frames read chunk positions c ^ (f & 3), all distinct.
//   Both are conflict-free (round 1's row-dependent swizzle on a 260-float stride measured 2x on the stores, 2.4x on the loads).
constexpr int kSlotStride = 272;

// Per-lane constants of the 8-lane layout (l = lane & 7, frame f = lane >> 3).
struct Rfft512Lane {
  int l;        // lane within the frame group
  int wr_off;   // pass-1 write offset (floats) inside a 16-float row
  int rd_a[4];  // pass-2 read offsets (floats) of column A chunks 0..3
  int rd_b[4];  // ... column B
  int ka_lo;    // kA of slot p (p < 8)  = ka_lo + 16 p
  int ka_hi;    // kA of slot p (p >= 8) = ka_hi + 16 p
  bool lane0;
};

__device__ __forceinline__ Rfft512Lane rfft512_lane_setup(int lane) {
  Rfft512Lane s;
  const int l = lane & 7;
  const int sw = (lane >> 3) & 3;  // chunk swizzle of this lane's frame
  s.l = l;
  s.lane0 = (l == 0);
  s.wr_off = ((((2 * l) >> 2) ^ sw) << 2) + ((2 * l) & 3);
  const int k1a = s.lane0 ? 8 : l;         // column A
  const int k1b = s.lane0 ? 0 : 16 - l;    // column B
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    s.rd_a[c] = k1a * 16 + ((c ^ sw) << 2);
    s.rd_b[c] = k1b * 16 + ((c ^ sw) << 2);
  }
  s.ka_lo = s.lane0 ? 8 : l;    // generic: l + 16p ; lane 0, p < 8: 8 + 16p
  s.ka_hi = s.lane0 ? 16 : l;   // lane 0, p >= 8: 16 (p + 1)
  return s;
}

// One output pair of the real-FFT split.  u = Z[kA], v = Z[256 - kA], (c, s) = (cos, sin)(2 pi kA / 512).
// Returns X[kA] in (xr, xi) and X[256 - kA] in (yr, yi).
__device__ __forceinline__ void rsplit_pair(float ur, float ui, float vr, float vi, float c, float s, float& xr,
                                            float& xi, float& yr, float& yi) {
  const float ar = ur + vr, ai = ui - vi;  // A = u + conj(v)
  const float br = ur - vr, bi = ui + vi;  // B = u - conj(v)
  const float tr = s * br - c * bi;        // T = i W B,  W = c - i s
  const float ti = c * br + s * bi;
  xr = ar - tr; xi = ai - ti;
  yr = ar + tr; yi = -(ai + ti);
}

// 512-point real FFT of 8 frames per wave.
//   In : (ar, ai)[m1] = windowed (x[32 m1 + 4l], x[32 m1 + 4l + 1]) * 1/2   (column m2 = 2l)
//        (br, bi)[m1] = windowed (x[32 m1 + 4l + 2], x[32 m1 + 4l + 3]) * 1/2 (column m2 = 2l + 1)
//   Out: for slot p = 0..15 (compile-time constant): emit(p, xr, xi, yr, yi) with (xr, xi) = X[kA(p)],
//        (yr, yi) = X[256 - kA(p)], kA(p) = (p < 8 ? ka_lo : ka_hi) + 16 p; emit128(re, im) = X[128],
//        meaningful on lane 0 only.  Emitting inside the loop keeps the live register set at the 64 data
//        registers instead of 128.
//   tw256: LDS float4 table [q*8 + l] = (W256^(q*2l), W256^(q*(2l+1))) as (re, im, re, im)
//   tw512: LDS float2 table [k] = (cos, sin)(2 pi k / 512), k = 0..256
//   slot : this frame's 256-float LDS area = tile base + frame * kSlotStride floats (16-byte aligned)
template <class Emit, class Emit128>
__device__ __forceinline__ void rfft512_x8(float (&ar)[16], float (&ai)[16], float (&br)[16], float (&bi)[16],
                                           const Rfft512Lane& s, const float4* __restrict__ tw256,
                                           const float2* __restrict__ tw512, float* __restrict__ slot,
                                           Emit&& emit, Emit128&& emit128) {
  // sched_barrier(0) at the phase seams: without them hipcc hoists the LDS table reads of the later phases
  // (3 x 60 registers) above the first FFT and the kernel drops from 3 to 2 waves per SIMD.
  __builtin_amdgcn_sched_barrier(0);
  fft16(ar, ai);
  fft16(br, bi);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int p = 0; p < 16; ++p) {
    const int q = rev4(p);
    if (q != 0) {
      const float4 w = tw256[q * 8 + s.l];
      cmul_inplace(ar[p], ai[p], w.x, w.y);
      cmul_inplace(br[p], bi[p], w.z, w.w);
    }
  }
  __builtin_amdgcn_sched_barrier(0);
  // ---- 16x16 transpose through LDS, real parts then imaginary parts -----------------------
#pragma unroll
  for (int p = 0; p < 16; ++p) {
    const int q = rev4(p);
    *reinterpret_cast<float2*>(slot + q * 16 + s.wr_off) = make_float2(ar[p], br[p]);
  }
  wave_lds_sync();
  float4 ca[4], cb[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    ca[c] = *reinterpret_cast<const float4*>(slot + s.rd_a[c]);
    cb[c] = *reinterpret_cast<const float4*>(slot + s.rd_b[c]);
  }
  wave_lds_sync();
#pragma unroll
  for (int p = 0; p < 16; ++p) {
    const int q = rev4(p);
    *reinterpret_cast<float2*>(slot + q * 16 + s.wr_off) = make_float2(ai[p], bi[p]);
  }
  wave_lds_sync();
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const float4 ia = *reinterpret_cast<const float4*>(slot + s.rd_a[c]);
    const float4 ib = *reinterpret_cast<const float4*>(slot + s.rd_b[c]);
    ar[4 * c] = ca[c].x; ar[4 * c + 1] = ca[c].y; ar[4 * c + 2] = ca[c].z; ar[4 * c + 3] = ca[c].w;
    br[4 * c] = cb[c].x; br[4 * c + 1] = cb[c].y; br[4 * c + 2] = cb[c].z; br[4 * c + 3] = cb[c].w;
    ai[4 * c] = ia.x; ai[4 * c + 1] = ia.y; ai[4 * c + 2] = ia.z; ai[4 * c + 3] = ia.w;
    bi[4 * c] = ib.x; bi[4 * c + 1] = ib.y; bi[4 * c + 2] = ib.z; bi[4 * c + 3] = ib.w;
  }
  wave_lds_sync();
  __builtin_amdgcn_sched_barrier(0);
  fft16(ar, ai);  // (ar, ai)[rev4(r)] = Z[k1a + 16 r]
  fft16(br, bi);  // (br, bi)[rev4(r)] = Z[k1b + 16 r]
  __builtin_amdgcn_sched_barrier(0);

  // ---- real-FFT split, 16 pair slots ---------------------------------------------------------
  // generic lane: slot p pairs a[p] with b[15-p].  lane 0 (a = column 8, b = column 0):
  //   p < 8 : a[p] with a[15-p]   (column 8 pairs with itself, r <-> 15 - r)
  //   p >= 8: b[p+1 mod 16] with b[15-p]   (column 0, r <-> 16 - r; p = 15 is the k = 0 / 256 self pair)
#pragma unroll
  for (int p = 0; p < 16; ++p) {
    float ur = ar[rev4(p)], ui = ai[rev4(p)];
    float vr = br[rev4(15 - p)], vi = bi[rev4(15 - p)];
    if (p < 8) {
      vr = s.lane0 ? ar[rev4(15 - p)] : vr;
      vi = s.lane0 ? ai[rev4(15 - p)] : vi;
    } else {
      ur = s.lane0 ? br[rev4((p + 1) & 15)] : ur;
      ui = s.lane0 ? bi[rev4((p + 1) & 15)] : ui;
    }
    const float2 w = tw512[(p < 8 ? s.ka_lo : s.ka_hi) + 16 * p];
    float xr, xi, yr, yi;
    rsplit_pair(ur, ui, vr, vi, w.x, w.y, xr, xi, yr, yi);
    emit(p, xr, xi, yr, yi);
  }
  // X[128] = conj(Z[128]) * 2 (window pre-scaled by 1/2): lane 0, column 0 (= b), r = 8
  emit128(2.0f * br[rev4(8)], -2.0f * bi[rev4(8)]);
}

}  // namespace ma
