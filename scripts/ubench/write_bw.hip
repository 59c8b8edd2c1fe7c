// write_bw.hip -- pure-store throughput of the rollout kernels' output pattern on gfx950: 1024 waves (one per SIMD),
// each writing one 704-byte chunk ([4 replicas][44 floats]) per "step", the chunks of one step contiguous over the
// waves, steps R*176 bytes apart:
//   mode 0  as k_rollout_pair does it: two dwordx2 stores per lane (speeds, positions), 88-byte runs;
//   mode 1  one dwordx4 per lane over the first 44 lanes: contiguous 16-byte pieces of the same 704 bytes;
//   mode 2  64 lanes x 16 B (1 KiB per wave and step).
// `spin` dependent v_fma_f32 per step stand in for the arithmetic between two stores.
//   hipcc --offload-arch=gfx950 -O3 -o write_bw write_bw.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef unsigned u2v __attribute__((__vector_size__(8)));
typedef unsigned u4v __attribute__((__vector_size__(16)));

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int R, int K, int spin) {
  const int lane = threadIdx.x & 63, wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int row = lane >> 4, kk = lane & 15;
  const unsigned rowb = 176u;
  const size_t step_bytes = MODE == 2 ? size_t(R / 4) * 1024 : size_t(R) * rowb;
  char* ob = reinterpret_cast<char*>(out);
  unsigned off_a, off_b = 0;
  if (MODE == 0) {
    const int k2 = kk < 11 ? kk : 10;
    off_a = unsigned(wave * 4 + row) * rowb + unsigned(k2) * 8u;
    off_b = off_a + 88u;
  } else if (MODE == 1) {
    off_a = lane < 44 ? unsigned(wave) * 704u + unsigned(lane) * 16u : 0xFFFFFF00u;
  } else {
    off_a = unsigned(wave) * 1024u + unsigned(lane) * 16u;
  }
  float acc = float(lane);
  for (int base = 0; base < K; base += 16) {
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(ob, 0, 0x40000000u, 0x00020000);   // the range check covers the VGPR offset only: idle lanes (0xFFFFFF00) are dropped
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      for (int q = 0; q < spin; ++q) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(acc));
      const unsigned so = unsigned(s) * unsigned(step_bytes);
      if (MODE == 0) {
        u2v d = {__builtin_bit_cast(unsigned, acc), unsigned(s)};
        __builtin_amdgcn_raw_buffer_store_b64(d, rs, off_a, so, 0);
        __builtin_amdgcn_raw_buffer_store_b64(d, rs, off_b, so, 0);
      } else {
        u4v d = {__builtin_bit_cast(unsigned, acc), unsigned(s), unsigned(lane), 7u};
        __builtin_amdgcn_raw_buffer_store_b128(d, rs, off_a, so, 0);
      }
    }
    ob += 16 * step_bytes;
  }
}
int main(int argc, char** argv) {
  const int R = 4096, K = 1504;
  float* out;
  CK(hipMalloc(&out, size_t(K) * (R / 4) * 1024 + (1 << 20)));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int spin : {0, 20, 40, 60, 80}) {
    for (int mode = 0; mode < 3; ++mode) {
      float best = 1e9;
      for (int it = 0; it < 6; ++it) {
        CK(hipEventRecord(e0, 0));
        if (mode == 0) k<0><<<256, 256>>>(out, R, K, spin);
        if (mode == 1) k<1><<<256, 256>>>(out, R, K, spin);
        if (mode == 2) k<2><<<256, 256>>>(out, R, K, spin);
        CK(hipEventRecord(e1, 0));
        CK(hipDeviceSynchronize());
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (it > 0 && ms < best) best = ms;
      }
      const double bytes = double(K) * (R / 4) * (mode == 2 ? 1024.0 : 704.0);
      printf("spin %2d mode %d: %.4f ms  %.0f GB/s\n", spin, mode, best, bytes / best / 1e6);
    }
  }
  return 0;
}
