// pair_bench.hip -- standalone timing harness of k_rollout_pair (C2 shape) for A/B experiments: compiles in
// seconds, variants through -D flags (FS_DIAG_* hooks in flowsim_pair.h).  Not a test: no result check.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 -I../../include -I../../flow_amd/csrc -o pair_bench pair_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
#include "flowsim_pair.h"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <typename T>
double run(int R, int N, int K, int block, int reps) {
  fs::DevView<T> s{};
  std::vector<T> pos(size_t(R) * N), vel(size_t(R) * N, T(0)), rl(R, T(230.0)), p(6 * N), len(N, T(5));
  for (int r = 0; r < R; ++r)
    for (int i = 0; i < N; ++i) pos[size_t(r) * N + i] = T(i * (100.0 / N + 5.0) + 0.3 * ((r * 7 + i * 3) % 11) / 11.0);
  const double pv[6] = {30, 1, 1, 1.5, 4, 2};
  for (int q = 0; q < 6; ++q) for (int i = 0; i < N; ++i) p[q * N + i] = T(pv[q]);
  T *dpos, *dvel, *drl, *dp, *dlen; int32_t* dtime; float *obs, *rew; uint8_t* done;
  CK(hipMalloc(&dpos, pos.size() * sizeof(T))); CK(hipMalloc(&dvel, pos.size() * sizeof(T)));
  CK(hipMalloc(&drl, R * sizeof(T))); CK(hipMalloc(&dp, p.size() * sizeof(T))); CK(hipMalloc(&dlen, N * sizeof(T)));
  CK(hipMalloc(&dtime, R * 4)); CK(hipMemset(dtime, 0, R * 4));
  CK(hipMalloc(&obs, size_t(K) * R * 2 * N * 4)); CK(hipMalloc(&rew, size_t(K) * R * 4)); CK(hipMalloc(&done, size_t(K) * R));
  CK(hipMemcpy(drl, rl.data(), R * sizeof(T), hipMemcpyHostToDevice));
  CK(hipMemcpy(dp, p.data(), p.size() * sizeof(T), hipMemcpyHostToDevice));
  CK(hipMemcpy(dlen, len.data(), N * sizeof(T), hipMemcpyHostToDevice));
  s.pos = dpos; s.vel = dvel; s.ring_len = drl; s.p = dp; s.length = dlen; s.time = dtime;
  s.R = R; s.N = N; s.dt = T(0.1); s.ramp = T(0.1 / 0.101); s.jlen = T(0.1); s.crash_gap = T(0); s.max_speed = T(30);
  s.target_velocity = T(10); s.max_cost = T(std::sqrt(double(N)) * 10.0); s.step_limit = 1500;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int waves = (R + 3) / 4, wpb = block / 64;
  double best = 1e9, sum = 0;
  for (int it = 0; it < reps + 2; ++it) {
    CK(hipMemcpy(dpos, pos.data(), pos.size() * sizeof(T), hipMemcpyHostToDevice));
    CK(hipMemcpy(dvel, vel.data(), vel.size() * sizeof(T), hipMemcpyHostToDevice));
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL((fs::k_rollout_pair<T, 16, true, true, false>), dim3((waves + wpb - 1) / wpb), dim3(block), 0, 0, s, K, obs, rew, done);
    CK(hipEventRecord(e1, 0));
    CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (it >= 2) { sum += ms; if (ms < best) best = ms; }
  }
  hipFree(dpos); hipFree(dvel); hipFree(drl); hipFree(dp); hipFree(dlen); hipFree(dtime); hipFree(obs); hipFree(rew); hipFree(done);
  return sum / reps;
}
int main(int argc, char** argv) {
  const int R = argc > 1 ? atoi(argv[1]) : 4096, K = argc > 2 ? atoi(argv[2]) : 1500, block = argc > 3 ? atoi(argv[3]) : 256;
  const double f = run<float>(R, 22, K, block, 8);
  const double m = run<double>(R, 22, K, block, 8);
  printf("R=%d K=%d block=%d  f32 %.4f ms (%.2f G)  mixed %.4f ms (%.2f G)\n", R, K, block, f, R * double(K) / f / 1e6, m, R * double(K) / m / 1e6);
  return 0;
}
