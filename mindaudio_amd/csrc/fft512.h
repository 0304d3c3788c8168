// 512-point real FFT on one 16-lane row of a wavefront (gfx950, wave64).
//
// A wave64 holds FOUR frames at once (one per 16-lane DPP row).  Per frame:
//   z[m] = x[2m] + i x[2m+1]            (256 complex points, 16 per lane, in VGPRs)
//   pass 1: 16-point FFT in registers over m1 (m = 16 m1 + lane)
//   twiddle W256^(lane*q), transpose 16x16 through a padded LDS slot (the ONLY exchange; re and im
//   go through the same 1 KiB slot one after the other)
//   pass 2: 16-point FFT in registers over m2  -> Z[lane + 16 k2]
//   real split: X[k] = E[k] + W512^k O[k] with the k <-> 256-k partner fetched from lane
//   (16-lane)%16 by two DPP row ops (row_mirror, row_ror:1) — no LDS, no bpermute.
// Result: lane j holds X[j + 16 k2], k2 = 0..15 (register slot rev4(k2)); lane 0 also X[256].
#pragma once
#include <hip/hip_runtime.h>

namespace ma {

typedef float v2f __attribute__((ext_vector_type(2)));

__device__ __forceinline__ v2f cmul(v2f a, v2f w) {
  return v2f{a.x * w.x - a.y * w.y, a.x * w.y + a.y * w.x};
}
__device__ __forceinline__ v2f mul_mi(v2f a) { return v2f{a.y, -a.x}; }  // * (-i)
__device__ __forceinline__ v2f mul_pi(v2f a) { return v2f{-a.y, a.x}; }  // * (+i)

// base-4 digit reversal of a 4-bit index: where X[k] lives after fft16
__host__ __device__ constexpr int rev4(int p) { return ((p >> 2) & 3) | ((p & 3) << 2); }

__device__ __forceinline__ void radix4(v2f& a0, v2f& a1, v2f& a2, v2f& a3) {
  v2f s0 = a0 + a2, s1 = a0 - a2, s2 = a1 + a3, s3 = a1 - a3;
  a0 = s0 + s2;
  a2 = s0 - s2;
  a1 = s1 + mul_mi(s3);
  a3 = s1 + mul_pi(s3);
}

// forward 16-point DFT, input natural order, output X[k] at a[rev4(k)]
__device__ __forceinline__ void fft16(v2f (&a)[16]) {
  constexpr float C1 = 0.92387953251128674f;  // cos(pi/8)
  constexpr float S1 = 0.38268343236508977f;  // sin(pi/8)
  constexpr float H = 0.70710678118654752f;
#pragma unroll
  for (int n2 = 0; n2 < 4; ++n2) radix4(a[n2], a[4 + n2], a[8 + n2], a[12 + n2]);
  // a[4*k1 + n2] *= W16^(n2*k1)
  a[5] = cmul(a[5], v2f{C1, -S1});
  a[6] = v2f{(a[6].x + a[6].y) * H, (a[6].y - a[6].x) * H};
  a[7] = cmul(a[7], v2f{S1, -C1});
  a[9] = v2f{(a[9].x + a[9].y) * H, (a[9].y - a[9].x) * H};
  a[10] = mul_mi(a[10]);
  a[11] = v2f{(a[11].y - a[11].x) * H, -(a[11].x + a[11].y) * H};
  a[13] = cmul(a[13], v2f{S1, -C1});
  a[14] = v2f{(a[14].y - a[14].x) * H, -(a[14].x + a[14].y) * H};
  a[15] = cmul(a[15], v2f{-C1, S1});
#pragma unroll
  for (int k1 = 0; k1 < 4; ++k1) radix4(a[4 * k1], a[4 * k1 + 1], a[4 * k1 + 2], a[4 * k1 + 3]);
}

// value of `v` held by lane (16 - j) % 16 of the same 16-lane row
__device__ __forceinline__ float row_partner(float v) {
  int x = __builtin_bit_cast(int, v);
  int m = __builtin_amdgcn_update_dpp(0, x, 0x140 /*row_mirror*/, 0xf, 0xf, false);  // m[j] = v[15-j]
  int r = __builtin_amdgcn_update_dpp(0, m, 0x121 /*row_ror:1*/, 0xf, 0xf, false);   // r[j] = m[(j-1)&15]
  return __builtin_bit_cast(float, r);
}

constexpr int kSlotFloats = 16 * 17;  // floats per frame slot (row stride 17: conflict-free both ways)

__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// In : a[m1] = windowed z[16*m1 + j] for this lane's frame (j = lane & 15)
// Out: a[rev4(k2)] = X[j + 16*k2]; returns X[256] (real) valid on lane j == 0.
//   tw256: LDS table [q*16 + j] = W256^(q*j);  tw512: LDS table [k] = (cos, sin)(2 pi k / 512)
//   slot : this frame's private LDS transpose area (kSlotFloats floats); the 16x16 complex transpose goes
//          through it one component at a time (re, then im), so the slot is only 1088 bytes and can alias
//          the frame's row of the power tile.
__device__ __forceinline__ float rfft512_row(v2f (&a)[16], int j, const v2f* __restrict__ tw256,
                                            const v2f* __restrict__ tw512, float* __restrict__ slot) {
  fft16(a);
#pragma unroll
  for (int p = 0; p < 16; ++p) {
    const int q = rev4(p);
    if (q != 0) a[p] = cmul(a[p], tw256[q * 16 + j]);
  }
#pragma unroll
  for (int p = 0; p < 16; ++p) slot[rev4(p) * 17 + j] = a[p].x;
  wave_lds_sync();
  float re[16];
#pragma unroll
  for (int m2 = 0; m2 < 16; ++m2) re[m2] = slot[j * 17 + m2];
  wave_lds_sync();
#pragma unroll
  for (int p = 0; p < 16; ++p) slot[rev4(p) * 17 + j] = a[p].y;
  wave_lds_sync();
#pragma unroll
  for (int m2 = 0; m2 < 16; ++m2) a[m2] = v2f{re[m2], slot[j * 17 + m2]};
  wave_lds_sync();
  fft16(a);  // a[rev4(k2)] = Z[j + 16 k2]

  const float x256 = a[0].x - a[0].y;  // lane 0: Z[0] -> X[256] = Re - Im
  v2f x[16];
#pragma unroll
  for (int k2 = 0; k2 < 16; ++k2) {
    const v2f z = a[rev4(k2)];
    const v2f src = a[rev4(15 - k2)];
    v2f zp = v2f{row_partner(src.x), row_partner(src.y)};  // Z[256-k] for j != 0
    const v2f own = a[rev4((16 - k2) & 15)];                // Z[256-k] for j == 0
    zp = (j == 0) ? own : zp;
    const v2f e = v2f{0.5f * (z.x + zp.x), 0.5f * (z.y - zp.y)};
    const v2f o = v2f{0.5f * (z.y + zp.y), -0.5f * (z.x - zp.x)};
    const v2f w = tw512[j + 16 * k2];  // (c, s); W512^k = c - i s
    x[k2] = v2f{e.x + (w.x * o.x + w.y * o.y), e.y + (w.x * o.y - w.y * o.x)};
  }
#pragma unroll
  for (int k2 = 0; k2 < 16; ++k2) a[rev4(k2)] = x[k2];
  return x256;
}

}  // namespace ma
