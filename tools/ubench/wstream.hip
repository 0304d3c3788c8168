// micro-benchmark: one workgroup of 4 waves per CU, every wave streams its own 1 KiB weight fragments from an L2-resident buffer and
// issues MFMAS_PER_FRAG MFMAs per fragment - the weight stream of ffn_packed / rows_packed / gemm_k256 - with the fragments going
//   MODE 0: L2 -> VGPR (global_load_dwordx4 ring, RING slots, counted vmcnt)
//   MODE 1: L2 -> LDS (global_load_lds_dwordx4 ring in LDS, RING slots) -> VGPR (ds_read_b128 one fragment ahead)
// hipcc --offload-arch=gfx950 -O3 tools/ubench/wstream.hip -o tools/ubench/wstream
#include <hip/hip_runtime.h>
#include <cstdio>
#include <type_traits>
#include <utility>
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void gl_void_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int... Is, class F>
__device__ __forceinline__ void sfor_impl(std::integer_sequence<int, Is...>, F&& f) { (f(std::integral_constant<int, Is>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void sfor(F&& f) { sfor_impl(std::make_integer_sequence<int, N>{}, f); }

template <int MODE, int RING, int NM>
__global__ __launch_bounds__(256, 1) void k(const char* buf, size_t bytes, int rounds, float* out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // wave w of every workgroup walks the same quarter of the buffer (as the packed weights: one consumer wave index per fragment)
  const size_t quarter = bytes / 4;
  const char* base = buf + wave * quarter;
  f32x4 acc_[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  bf16x8 bfr;
  for (int i = 0; i < 8; ++i) bfr[i] = (__bf16)1.0f;
  char* ring_lds = smem + wave * (RING * 1024);
  const uint32_t lds_rd = (uint32_t)(uintptr_t)(lds_void_t*)(ring_lds) + lane * 16;
  size_t off = 0;  // next fragment to request (bytes into the quarter)
  auto next_src = [&]() {
    const char* s = base + off + lane * 16;
    off += 1024;
    if (off >= quarter) off = 0;
    return s;
  };
  if constexpr (MODE == 0) {
    bf16x8 ring_[RING];
    sfor<RING>([&](auto qc) {
      constexpr int q = decltype(qc)::value;
      auto& ring = ring_;
      const char* s = next_src();
      asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(ring[q]) : "v"(s) : "memory");
    });
    for (int r = 0; r < rounds; ++r) {
      sfor<RING>([&](auto qc) {
        constexpr int q = decltype(qc)::value;
        auto& ring = ring_;
        auto& acc = acc_;
        asm volatile("s_waitcnt vmcnt(%1)" : "+v"(ring[q]) : "n"(RING - 1) : "memory");
        __builtin_amdgcn_sched_barrier(0);
        sfor<NM>([&](auto mc) {
          constexpr int m = decltype(mc)::value;
          acc[m & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ring[q], bfr, acc[m & 3], 0, 0, 0);
        });
        __builtin_amdgcn_sched_barrier(0);
        const char* s = next_src();
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(ring[q]) : "v"(s) : "memory");
      });
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    sfor<RING>([&](auto qc) { constexpr int q = decltype(qc)::value; auto& ring = ring_; asm volatile("" : "+v"(ring[q])); });
  } else {
    sfor<RING>([&](auto qc) {
      constexpr int q = decltype(qc)::value;
      __builtin_amdgcn_global_load_lds((gl_void_t*)next_src(), (lds_void_t*)(ring_lds + q * 1024), 16, 0, 0);
    });
    bf16x8 cur_, nxt_;
    auto& cur = cur_;
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(RING - 1) : "memory");
    asm volatile("ds_read_b128 %0, %1" : "=v"(cur) : "v"(lds_rd) : "memory");
    for (int r = 0; r < rounds; ++r) {
      sfor<RING>([&](auto qc) {
        constexpr int q = decltype(qc)::value;
        constexpr int qn = (q + 1) % RING;
        auto& acc = acc_;
        auto& cur = cur_;
        auto& nxt = nxt_;
        // slot q + 1 has landed <=> at most RING - 2 younger loads outstanding (slot q's refill has not been issued yet)
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(RING - 2) : "memory");
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(nxt) : "v"(lds_rd), "n"(qn * 1024) : "memory");
        asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(cur)::"memory");
        __builtin_amdgcn_sched_barrier(0);
        sfor<NM>([&](auto mc) {
          constexpr int m = decltype(mc)::value;
          acc[m & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cur, bfr, acc[m & 3], 0, 0, 0);
        });
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_global_load_lds((gl_void_t*)next_src(), (lds_void_t*)(ring_lds + q * 1024), 16, 0, 0);
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(nxt)::"memory");
        cur = nxt;
      });
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i) s += acc_[i][0] + acc_[i][1] + acc_[i][2] + acc_[i][3];
  if (s == 12345.f) out[tid] = s;
}

int main() {
  char* buf;
  float* out;
  const size_t bytes = 2u << 20;  // 2 MiB: L2-resident, 512 KiB per wave index
  (void)hipMalloc(&buf, bytes);
  (void)hipMalloc(&out, 4096);
  (void)hipMemset(buf, 0, bytes);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  auto run = [&](const char* nm, auto kern, int ring, int nm_, size_t lds) {
    const int rounds = 4096 / ring;  // 4096 fragments = 4 MiB per wave
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    float ms = 0;
    for (int r = 0; r < 3; ++r) {
      (void)hipEventRecord(e0);
      kern<<<256, 256, lds>>>(buf, bytes, rounds, out);
      (void)hipEventRecord(e1);
      (void)hipEventSynchronize(e1);
      (void)hipEventElapsedTime(&ms, e0, e1);
    }
    const double per_wg = 4.0 * rounds * ring * 1024;
    printf("%-40s ring %2d, %d MFMA/fragment: %.3f ms, %.1f GB/s per CU, MFMA-bound time %.3f ms\n", nm, ring, nm_, ms,
           per_wg / (ms * 1e-3) / 1e9, rounds * ring * nm_ * 16 / 2.4e6);
  };
  run("L2 -> VGPR", k<0, 16, 4>, 16, 4, 0);
  run("L2 -> VGPR", k<0, 24, 4>, 24, 4, 0);
  run("L2 -> VGPR", k<0, 16, 3>, 16, 3, 0);
  run("L2 -> VGPR", k<0, 16, 2>, 16, 2, 0);
  run("L2 -> LDS -> VGPR", k<1, 16, 4>, 16, 4, 4 * 16 * 1024);
  run("L2 -> LDS -> VGPR", k<1, 24, 4>, 24, 4, 4 * 24 * 1024);
  run("L2 -> LDS -> VGPR", k<1, 32, 4>, 32, 4, 4 * 32 * 1024);
  run("L2 -> LDS -> VGPR", k<1, 16, 3>, 16, 3, 4 * 16 * 1024);
  run("L2 -> LDS -> VGPR", k<1, 16, 2>, 16, 2, 4 * 16 * 1024);
  run("L2 -> LDS -> VGPR", k<1, 32, 2>, 32, 2, 4 * 32 * 1024);
  return 0;
}
