// flowsim_policy.h -- the closed loop in ONE kernel: policy -> action -> Env.step, K times, for the reference's RL ring
// experiment (examples/train.py:110-212 is what it replaces: one Python call + socket round trips per step and
// environment; examples/exp_configs/rl/singleagent/singleagent_ring.py: 21 IDM + 1 RL vehicle, WaveAttenuationPOEnv).
//
// What a learner gets per fragment: obs [K+1, R, 3], actions [K, R, 1], log-probabilities [K, R], rewards, done flags --
// the rollout batch of a policy-gradient method -- with the state of a replica in registers for the whole fragment and the
// policy evaluated where the observation is produced.  The policy is the reference's default model class: a fully
// connected network of up to three hidden layers of 32 tanh units (examples/train.py:152 `fcnet_hiddens [32, 32, 32]`)
// with a diagonal Gaussian head -- either two outputs (mean, log std: RLlib's default) or one output and a free log std.
//
// Mapping: the simulator's (k_ring_pair, flowsim_ringrl.h): a row of 16 lanes = one replica, four replicas per wave.
// Lane j of a row holds hidden units j and j + 16; a layer's inputs are fetched from the row with DPP row broadcasts
// (folded into the FMAs), its weights are read from an LDS copy (16 b128 reads per lane and layer), the two outputs are
// reduced over the row with the xor-tree of seg_sum.  The sampled action uses the replica's own Philox stream (keyed by
// the policy's seed, the global replica index and a per-replica counter that a fragment continues where the last one
// stopped).  policy_eval is ONE device function shared with k_policy_act (the eager form: fs_policy_act_dev followed by
// fs_step_dev), so the fragment is bit-identical to eager stepping by construction (tests/test_policy_gpu.py).
// Every operation is an explicit fma / hardware exp2 / rcp: the arithmetic is defined by this file (a torch module with
// the same weights agrees to ~1e-6, not bit for bit).
#pragma once
#include "flowsim_ringrl.h"

namespace fs {

struct PolicyView {
  const float* w;          // device: per layer W [out][in] row-major, then b [out]
  const float* log_std;    // device [1], or NULL: the network's second output is the log std
  uint32_t* ctr;           // device [R]: actions sampled so far per replica (the draw counter of the policy's stream)
  int in_dim, num_hidden, n_out;
  uint32_t seed_lo, seed_hi;
};

struct alignas(16) PolicyLds {
  // hidden layers 2 and 3: [layer][quad q][lane j][4] = (W[j][2q], W[j+16][2q], W[j][2q+1], W[j+16][2q+1]) -- a packed FMA per
  // input, and the 16 lanes of a row read 256 consecutive bytes per quad (a per-lane row of 256 B would put all of them on
  // the same LDS banks: a 16-way conflict on every read, measured 1.4 -> 0.9 G)
  float w_hid[2][16][16][4];
  float w_in[32][4];       // layer 1: [unit][input (3, padded)]
  // layer 1 of a WIDE observation (in_dim = 2 * half <= 32, e.g. AccelEnv's speeds and positions): input i < half is the
  // first value of lane i, input half + i its second value -- the layout and the summation order of a hidden layer
  float w_wide[16][16][4];
  float b[3][32];
  float w_out[2][32];
  float b_out[2];
  float obs[4][4][4];      // [wave of the block][row][value]: the observation of a replica, handed from its RL lane to the row
};

__device__ __forceinline__ void policy_load(const PolicyView& pv, PolicyLds* L, int tid, int nthreads) {
  // weights -> LDS (once per launch); layout of pv.w: [W1 32x3][b1 32][W2 32x32][b2][W3 32x32][b3][Wout n_out x 32][bout]
  const float* p = pv.w;
  const bool wide = pv.in_dim > 4;
  for (int e = tid; e < 32 * 4; e += nthreads) L->w_in[e / 4][e % 4] = (!wide && (e % 4) < pv.in_dim) ? p[(e / 4) * pv.in_dim + (e % 4)] : 0.0f;
  {
    const int half = pv.in_dim >> 1;
    for (int e = tid; e < 32 * 32; e += nthreads) {
      const int u = e / 32, ip = e % 32;                       // ip: place of the input in the hidden-layer order
      const int i = ip < 16 ? ip : half + (ip - 16);           // ... and its index in the observation
      const bool used = wide && (ip & 15) < half;
      L->w_wide[ip >> 1][u & 15][(ip & 1) * 2 + (u >> 4)] = used ? p[u * pv.in_dim + i] : 0.0f;
    }
  }
  p += 32 * pv.in_dim;
  for (int e = tid; e < 32; e += nthreads) L->b[0][e] = p[e];
  p += 32;
  for (int l = 0; l < 2; ++l) {
    const bool have = l + 1 < pv.num_hidden;
    for (int e = tid; e < 32 * 32; e += nthreads) {
      const int u = e / 32, i = e % 32;
      L->w_hid[l][i >> 1][u & 15][(i & 1) * 2 + (u >> 4)] = have ? p[e] : 0.0f;
    }
    if (have) p += 32 * 32;
    for (int e = tid; e < 32; e += nthreads) L->b[l + 1][e] = have ? p[e] : 0.0f;
    if (have) p += 32;
  }
  for (int e = tid; e < 2 * 32; e += nthreads) L->w_out[e / 32][e % 32] = (e / 32) < pv.n_out ? p[e] : 0.0f;
  p += pv.n_out * 32;
  if (tid < 2) L->b_out[tid] = tid < pv.n_out ? p[tid] : 0.0f;
  __syncthreads();
}

// (a, b, c) of lanes (k, k + 1, k + 2) mod 16 of every 16-lane row -> all lanes of the row (k wave-uniform): three DPP row
// broadcasts behind one jump on k, where three ds_bpermute would be an LDS round trip the wave waits out (nothing else
// to issue: the policy's first layer needs the three values)
__device__ __forceinline__ void row_bcast3(int k, float a, float b, float c, float& oa, float& ob, float& oc) {
#define FS_RB3(K_) case K_: oa = dpp<DPP_ROW_NEWBCAST0 + K_>(a); ob = dpp<DPP_ROW_NEWBCAST0 + ((K_ + 1) & 15)>(b); \
                           oc = dpp<DPP_ROW_NEWBCAST0 + ((K_ + 2) & 15)>(c); break;
  switch (k & 15) {
    FS_RB3(0) FS_RB3(1) FS_RB3(2) FS_RB3(3) FS_RB3(4) FS_RB3(5) FS_RB3(6) FS_RB3(7)
    FS_RB3(8) FS_RB3(9) FS_RB3(10) FS_RB3(11) FS_RB3(12) FS_RB3(13) FS_RB3(14) FS_RB3(15)
  }
#undef FS_RB3
}
// the same lane k three times
__device__ __forceinline__ void row_bcast1x3(int k, float a, float b, float c, float& oa, float& ob, float& oc) {
#define FS_RB1(K_) case K_: oa = dpp<DPP_ROW_NEWBCAST0 + K_>(a); ob = dpp<DPP_ROW_NEWBCAST0 + K_>(b); \
                           oc = dpp<DPP_ROW_NEWBCAST0 + K_>(c); break;
  switch (k & 15) {
    FS_RB1(0) FS_RB1(1) FS_RB1(2) FS_RB1(3) FS_RB1(4) FS_RB1(5) FS_RB1(6) FS_RB1(7)
    FS_RB1(8) FS_RB1(9) FS_RB1(10) FS_RB1(11) FS_RB1(12) FS_RB1(13) FS_RB1(14) FS_RB1(15)
  }
#undef FS_RB1
}

__device__ __forceinline__ float policy_tanh(float z) {
  const float e = __builtin_amdgcn_exp2f(z * 2.885390081777927f);       // exp(2 z)
  const float r = __builtin_amdgcn_rcpf(e + 1.0f);
  return __builtin_fmaf(-2.0f, r, 1.0f);                                // 1 - 2 / (exp(2 z) + 1)
}

// mean and log std of the action distribution for the observation (o0, o1, o2) of THIS row's replica (every lane of the
// row passes the same three values); j = lane within the row
// WIDE: the observation is two values per lane (o0: input j, o1: input half + j of THIS lane; o2 unused)
template <int ROW, bool WIDE = false>
__device__ __forceinline__ void policy_eval(const PolicyView& pv, const PolicyLds* L, int j, float o0, float o1, float o2,
                                            float& mu, float& log_std) {
  static_assert(ROW == 16, "policy_eval: a row of 16 lanes holds the 32 units of a layer");
  // The first two weight quads of a hidden layer are read one layer AHEAD (before the previous layer's tanh): the layer's
  // FMAs start on them while its other fourteen reads are in flight, instead of waiting out an LDS round trip at the top
  // of every layer.  (All sixteen ahead was measured slower: the 64 registers stay live through tanh and the kernel spills.)
  float4 wa0 = make_float4(0.0f, 0.0f, 0.0f, 0.0f), wa1 = wa0;
  auto load_ahead = [&](int l) {
    wa0 = *reinterpret_cast<const float4*>(L->w_hid[l][0][j]);
    wa1 = *reinterpret_cast<const float4*>(L->w_hid[l][1][j]);
  };
  if (pv.num_hidden > 1) load_ahead(0);
  // layer 1
  float ha, hb;      // units j and j + 16
  if constexpr (WIDE) {
    // the hidden layers' form (below): four independent accumulators, inputs by row broadcasts folded into packed FMAs
    float4 ww[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) ww[q] = *reinterpret_cast<const float4*>(L->w_wide[q][j]);
    f2 z0 = {L->b[0][j], L->b[0][j + 16]}, z1 = {0.0f, 0.0f}, z2 = {0.0f, 0.0f}, z3 = {0.0f, 0.0f};
    static_for<4>([&](auto q_c) {
      constexpr int q = decltype(q_c)::value;
      const float a0 = dpp<DPP_ROW_NEWBCAST0 + 4 * q>(o0), a1 = dpp<DPP_ROW_NEWBCAST0 + 4 * q + 1>(o0);
      const float a2 = dpp<DPP_ROW_NEWBCAST0 + 4 * q + 2>(o0), a3 = dpp<DPP_ROW_NEWBCAST0 + 4 * q + 3>(o0);
      const float b0 = dpp<DPP_ROW_NEWBCAST0 + 4 * q>(o1), b1 = dpp<DPP_ROW_NEWBCAST0 + 4 * q + 1>(o1);
      const float b2 = dpp<DPP_ROW_NEWBCAST0 + 4 * q + 2>(o1), b3 = dpp<DPP_ROW_NEWBCAST0 + 4 * q + 3>(o1);
      z0 = fma2(f2{ww[2 * q].x, ww[2 * q].y}, splat(a0), z0);
      z1 = fma2(f2{ww[2 * q].z, ww[2 * q].w}, splat(a1), z1);
      z2 = fma2(f2{ww[2 * q + 1].x, ww[2 * q + 1].y}, splat(a2), z2);
      z3 = fma2(f2{ww[2 * q + 1].z, ww[2 * q + 1].w}, splat(a3), z3);
      z0 = fma2(f2{ww[8 + 2 * q].x, ww[8 + 2 * q].y}, splat(b0), z0);
      z1 = fma2(f2{ww[8 + 2 * q].z, ww[8 + 2 * q].w}, splat(b1), z1);
      z2 = fma2(f2{ww[8 + 2 * q + 1].x, ww[8 + 2 * q + 1].y}, splat(b2), z2);
      z3 = fma2(f2{ww[8 + 2 * q + 1].z, ww[8 + 2 * q + 1].w}, splat(b3), z3);
    });
    const f2 z = pk_add(pk_add(z0, z1), pk_add(z2, z3));
    ha = policy_tanh(z.x);
    hb = policy_tanh(z.y);
  } else {
    const float4 wa = *reinterpret_cast<const float4*>(L->w_in[j]), wb = *reinterpret_cast<const float4*>(L->w_in[j + 16]);
    float za = L->b[0][j], zb = L->b[0][j + 16];
    za = __builtin_fmaf(wa.x, o0, za); zb = __builtin_fmaf(wb.x, o0, zb);
    za = __builtin_fmaf(wa.y, o1, za); zb = __builtin_fmaf(wb.y, o1, zb);
    za = __builtin_fmaf(wa.z, o2, za); zb = __builtin_fmaf(wb.z, o2, zb);
    ha = policy_tanh(za);
    hb = policy_tanh(zb);
  }
  // hidden layers 2 .. num_hidden: z[u] = b[u] + sum_i W[u][i] h[i], i ascending; input i < 16 sits in `ha` of lane i,
  // input i >= 16 in `hb` of lane i - 16 (row broadcasts, folded into the FMAs)
#pragma unroll 1
  for (int l = 0; l + 1 < pv.num_hidden; ++l) {
    // all sixteen weight quads of the layer first (the LDS reads in flight together), then four independent accumulators
    // (inputs i with the same i mod 4 share one; the chain of dependent packed FMAs is 8 long instead of 32), combined
    // as ((b + z0) + z1) + (z2 + z3)
    float4 ww[16];                                 // ww[i / 2] = weights of inputs i, i + 1
    ww[0] = wa0;
    ww[1] = wa1;
#pragma unroll
    for (int q = 2; q < 16; ++q) ww[q] = *reinterpret_cast<const float4*>(L->w_hid[l][q][j]);
    f2 z0 = {L->b[l + 1][j], L->b[l + 1][j + 16]}, z1 = {0.0f, 0.0f}, z2 = {0.0f, 0.0f}, z3 = {0.0f, 0.0f};
    static_for<4>([&](auto q_c) {                  // inputs 4q .. 4q + 3 (`ha` of lanes 4q ..) and 16 + 4q .. (`hb`)
      constexpr int q = decltype(q_c)::value;
      const float a0 = dpp<DPP_ROW_NEWBCAST0 + 4 * q>(ha), a1 = dpp<DPP_ROW_NEWBCAST0 + 4 * q + 1>(ha);
      const float a2 = dpp<DPP_ROW_NEWBCAST0 + 4 * q + 2>(ha), a3 = dpp<DPP_ROW_NEWBCAST0 + 4 * q + 3>(ha);
      const float b0 = dpp<DPP_ROW_NEWBCAST0 + 4 * q>(hb), b1 = dpp<DPP_ROW_NEWBCAST0 + 4 * q + 1>(hb);
      const float b2 = dpp<DPP_ROW_NEWBCAST0 + 4 * q + 2>(hb), b3 = dpp<DPP_ROW_NEWBCAST0 + 4 * q + 3>(hb);
      z0 = fma2(f2{ww[2 * q].x, ww[2 * q].y}, splat(a0), z0);
      z1 = fma2(f2{ww[2 * q].z, ww[2 * q].w}, splat(a1), z1);
      z2 = fma2(f2{ww[2 * q + 1].x, ww[2 * q + 1].y}, splat(a2), z2);
      z3 = fma2(f2{ww[2 * q + 1].z, ww[2 * q + 1].w}, splat(a3), z3);
      z0 = fma2(f2{ww[8 + 2 * q].x, ww[8 + 2 * q].y}, splat(b0), z0);
      z1 = fma2(f2{ww[8 + 2 * q].z, ww[8 + 2 * q].w}, splat(b1), z1);
      z2 = fma2(f2{ww[8 + 2 * q + 1].x, ww[8 + 2 * q + 1].y}, splat(b2), z2);
      z3 = fma2(f2{ww[8 + 2 * q + 1].z, ww[8 + 2 * q + 1].w}, splat(b3), z3);
    });
    if (l + 2 < pv.num_hidden) load_ahead(l + 1);  // (wave-uniform)
    const f2 z = pk_add(pk_add(z0, z1), pk_add(z2, z3));
    const float za = z.x, zb = z.y;
    ha = policy_tanh(za);
    hb = policy_tanh(zb);
  }
  // head: two outputs, each the row's tree sum of the lanes' two products
  float p0 = L->w_out[0][j] * ha, p1 = L->w_out[1][j] * ha;
  p0 = __builtin_fmaf(L->w_out[0][j + 16], hb, p0);
  p1 = __builtin_fmaf(L->w_out[1][j + 16], hb, p1);
  mu = seg_sum<ROW>(p0) + L->b_out[0];
  const float o1_ = seg_sum<ROW>(p1) + L->b_out[1];
  log_std = pv.log_std != nullptr ? pv.log_std[0] : o1_;
}

// the action: mean + std * g, g the replica's next standard normal draw (Philox block of four per counter / 4, column
// 0x40000000 + 0: a stream of its own next to the vehicles' noise); log-probability of a 1-d diagonal Gaussian
__device__ __forceinline__ void policy_sample(const PolicyView& pv, uint32_t replica, uint32_t ctr, float mu, float log_std,
                                              float& action, float& logp, NoiseBlock<float>* nzb = nullptr) {
  // (a fragment keeps the four draws of a block over four steps: NoiseBlock::draw is gauss() bit for bit)
  const float g = nzb ? nzb->draw(pv.seed_lo, pv.seed_hi, replica, 0x40000000u, ctr)
                      : gauss<float>(pv.seed_lo, pv.seed_hi, replica, 0x40000000u, ctr);
  const float sd = __builtin_amdgcn_exp2f(log_std * 1.4426950408889634f);
  action = __builtin_fmaf(sd, g, mu);
  logp = __builtin_fmaf(-0.5f * g, g, -log_std) - 0.9189385332046727f;
}

// eager form: actions and log-probabilities for the observations obs [R, 3]; advances the sampling counters
template <int ROW>     // (a template so that every object of the library may include this header)
__global__ __launch_bounds__(256) void k_policy_act(PolicyView pv, int R, uint32_t rep0, const float* __restrict__ obs,
                                                    float* __restrict__ act, float* __restrict__ logp) {
  __shared__ PolicyLds L;
  policy_load(pv, &L, threadIdx.x, blockDim.x);
  const int lane = threadIdx.x & 63, j = lane & 15;
  const int r = (blockIdx.x * blockDim.x + threadIdx.x) >> 4;
  const int rr = r < R ? r : R - 1;
  const float* o = obs + size_t(rr) * pv.in_dim;
  float mu, ls;
  if (pv.in_dim > 4) {                                   // (wave-uniform) two values per lane: inputs j and half + j
    const int half = pv.in_dim >> 1;
    const float ia = j < half ? o[j] : 0.0f, ib = j < half ? o[half + j] : 0.0f;
    policy_eval<ROW, true>(pv, &L, j, ia, ib, 0.0f, mu, ls);
  } else {
    policy_eval<ROW>(pv, &L, j, o[0], pv.in_dim > 1 ? o[1] : 0.0f, pv.in_dim > 2 ? o[2] : 0.0f, mu, ls);
  }
  const uint32_t c = pv.ctr[rr];
  float a, lp;
  policy_sample(pv, rep0 + uint32_t(rr), c, mu, ls, a, lp);
  if (r < R && j == 0) {
    act[r] = a;
    logp[r] = lp;
    pv.ctr[r] = c + 1u;
  }
}

// K x (policy -> action -> Env.step [-> reset of a finished episode]) for rings of IDM vehicles and ONE RL vehicle with
// the WaveAttenuationPOEnv head.  obs [K+1, R, 3] (obs[0]: the observation of the state the fragment starts from),
// act [K, R], logp [K, R], rew [K, R], done [K, R].  The simulator part is k_ring_pair's arithmetic statement by statement.
template <typename T, bool NOISE, bool FAST>
__global__ __launch_bounds__(256) void k_ring_policy(DevView<T> s, PolicyView pv, int num_steps, int reset_done,
                                                     int warmup_steps, float* __restrict__ obs, float* __restrict__ act,
                                                     float* __restrict__ logp, float* __restrict__ rew,
                                                     uint8_t* __restrict__ done) {
  constexpr int ROW = 16;
  constexpr bool MIXED = sizeof(T) == 8;
  constexpr int RPW = 64 / ROW;
  __shared__ PolicyLds PL;
  policy_load(pv, &PL, threadIdx.x, blockDim.x);
  const int lane = threadIdx.x & 63;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int wib = (threadIdx.x >> 6) & 3;
  const int row = lane / ROW;
  const int k = lane % ROW;
  const int r = wave * RPW + row;
  const int N = s.N;
  const int LP = N >> 1;
  const bool rvalid = r < s.R;
  const bool valid = rvalid && k < LP;
  const int rr = rvalid ? r : s.R - 1;
  const int kk = k < LP ? k : LP - 1;
  const bool last = (kk == LP - 1);
  const int iA = 2 * kk, iB = iA + 1;
  const size_t idx = size_t(rr) * N + iA;

  const bool rlA = s.ctrl[iA] == FS_CTRL_RL, rlB = s.ctrl[iB] == FS_CTRL_RL;
  f2 p[6];
#pragma unroll
  for (int q = 0; q < 6; ++q) p[q] = f2{float(s.p[q * N + iA]), float(s.p[q * N + iB])};
  {
    const float dflt[6] = {30.0f, 1.0f, 1.0f, 1.5f, 4.0f, 2.0f};
#pragma unroll
    for (int q = 0; q < 6; ++q) {
      p[q].x = rlA ? dflt[q] : p[q].x;
      p[q].y = rlB ? dflt[q] : p[q].y;
    }
  }
  const T lenB = s.length[iB];
  const T len_nextA = next_a<ROW>(T(s.length[iA]), last, lane);
  T L = s.ring_len[rr] + T(4) * s.jlen;
  int tcount = s.time[rr];
  uint32_t nctr = NOISE ? s.noise_ctr[rr] : 0u;
  uint32_t pctr = pv.ctr[rr];
  const f2 sigma = NOISE ? f2{float(s.noise[iA]), float(s.noise[iB])} : f2{0.0f, 0.0f};
  const bool noisyA = NOISE && sigma.x > 0.0f && !rlA, noisyB = NOISE && sigma.y > 0.0f && !rlB;
  float gA[4] = {0.0f, 0.0f, 0.0f, 0.0f}, gB[4] = {0.0f, 0.0f, 0.0f, 0.0f};
  if constexpr (NOISE) {
    if ((nctr & 3u) != 0u) {
      gauss4<float>(s.seed_lo, s.seed_hi, s.rep0 + uint32_t(rr), uint32_t(iA), nctr >> 2, gA, s.noise_exact != 0);
      gauss4<float>(s.seed_lo, s.seed_hi, s.rep0 + uint32_t(rr), uint32_t(iB), nctr >> 2, gB, s.noise_exact != 0);
      for (uint32_t q = 0; q < (nctr & 3u); ++q) {
        gA[0] = gA[1]; gA[1] = gA[2]; gA[2] = gA[3];
        gB[0] = gB[1]; gB[1] = gB[2]; gB[2] = gB[3];
      }
    }
  }
  const float dt = float(s.dt), ramp = float(s.ramp);
  const f2 two_sqrt_ab = {2.0f * tsqrt(p[2].x * p[3].x), 2.0f * tsqrt(p[2].y * p[3].y)};
  const f2 rc_v0 = {1.0f / p[0].x, 1.0f / p[0].y}, rc_ab = {1.0f / two_sqrt_ab.x, 1.0f / two_sqrt_ab.y};
  const f2 one = splat(1.0f), dt2 = splat(dt), ramp2 = splat(ramp);
  const f2 gap2 = splat(float(s.crash_gap));
  const double dt64 = double(s.dt), ramp64 = double(s.ramp);
  const f2 len_lead = {float(lenB), float(len_nextA)};
  // the loop length and what depends on it: a reset inside the fragment takes the replica's pending length
  f2 L2 = splat(float(L));
  double L64 = double(L);
  SumoPair sc;
  double floor0A, floor0B, adtA, adtB, ddtA, ddtB;
  {
    const int mA = s.speed_mode[iA], mB = s.speed_mode[iB];
    const float maA = float(s.max_accel[iA]), maB = float(s.max_accel[iB]);
    const float mdA = float(s.max_decel[iA]), mdB = float(s.max_decel[iB]);
    sc.tau = f2{float(s.sumo_tau[iA]), float(s.sumo_tau[iB])};
    sc.min_gap = f2{float(s.sumo_min_gap[iA]), float(s.sumo_min_gap[iB])};
    sc.maxa = f2{maA, maB};
    sc.smax = f2{float(s.sumo_max_speed[iA]), float(s.sumo_max_speed[iB])};
    sc.rc_smax = f2{1.0f / sc.smax.x, 1.0f / sc.smax.y};
    sc.ts = f2{2.0f * tsqrt(maA * mdA), 2.0f * tsqrt(maB * mdB)};
    sc.rc_ts = f2{1.0f / sc.ts.x, 1.0f / sc.ts.y};
    const float BIG = 3.0e38f;
    sc.floor0 = f2{(mA & 1) ? 0.0f : BIG, (mB & 1) ? 0.0f : BIG};
    sc.adt = f2{(mA & 2) ? maA * dt : BIG, (mB & 2) ? maB * dt : BIG};
    sc.ddt = f2{(mA & 4) ? mdA * dt : BIG, (mB & 4) ? mdB * dt : BIG};
    floor0A = double(sc.floor0.x); floor0B = double(sc.floor0.y);
    adtA = (mA & 2) ? double(s.max_accel[iA]) * dt64 : double(BIG);
    adtB = (mB & 2) ? double(s.max_accel[iB]) * dt64 : double(BIG);
    ddtA = (mA & 4) ? double(s.max_decel[iA]) * dt64 : double(BIG);
    ddtB = (mB & 4) ? double(s.max_decel[iB]) * dt64 : double(BIG);
  }

  f2 x, v;
  double xdA = 0, xdB = 0, vdA = 0, vdB = 0;
  auto load_state = [&](const T* px, const T* pvv) {
    if (MIXED) {
      xdA = double(px[idx]); xdB = double(px[idx + 1]);
      vdA = double(pvv[idx]); vdB = double(pvv[idx + 1]);
      v = f2{float(vdA), float(vdB)};
      x = f2{0.0f, 0.0f};
    } else {
      x = f2{float(px[idx]), float(px[idx + 1])};
      v = f2{float(pvv[idx]), float(pvv[idx + 1])};
    }
  };
  load_state(s.pos, s.vel);
  f2 dgap = {0.0f, 0.0f};
  double dgA = 0, dgB = 0;
  auto headway = [&]() -> f2 {
    if (MIXED) {
      const double xn = next_a<ROW>(xdA, last, lane);
      double dA = xdB - xdA, dB = xn - xdB;
      dA = dA < 0.0 ? dA + L64 : dA;
      dB = dB < 0.0 ? dB + L64 : dB;
      dgA = dA;
      dgB = dB;
      return f2{float(dA - double(lenB)), float(dB - double(len_nextA))};
    } else {
      const f2 xl = {x.y, next_a<ROW>(x.x, last, lane)};
      f2 d = pk_sub(xl, x);
      const f2 dw = pk_add(d, L2);
      d.x = nonneg_else(d.x, dw.x);
      d.y = nonneg_else(d.y, dw.y);
      dgap = d;
      return pk_sub(d, len_lead);
    }
  };
  f2 h = headway();
  f2 vl = {v.y, next_a<ROW>(v.x, last, lane)};

  // (k_ring_pair's forms: an unconditional clamp with +-3e38 bounds without clip_actions; one-instruction min / max
  // evaluated before the selects that take them)
  const bool clip_on = s.clip_actions != 0;
  const float act_lo = clip_on ? float(s.act_lo) : -3.0e38f, act_hi = clip_on ? float(s.act_hi) : 3.0e38f;
  auto clip = [&](float a) -> float { return hmin(hmax(a, act_lo), act_hi); };
  auto noise_term = [&](bool live) -> f2 {
    f2 nz = {-0.0f, -0.0f};
    if constexpr (NOISE) {
      const bool fresh = live && (nctr & 3u) == 0u;
      if (__ballot(fresh) != 0ull) {
        if (fresh) {
          gauss4<float>(s.seed_lo, s.seed_hi, s.rep0 + uint32_t(rr), uint32_t(iA), nctr >> 2, gA, s.noise_exact != 0);
          gauss4<float>(s.seed_lo, s.seed_hi, s.rep0 + uint32_t(rr), uint32_t(iB), nctr >> 2, gB, s.noise_exact != 0);
        }
      }
      const float tA = sigma.x * gA[0], tB = sigma.y * gB[0];
      nz.x = noisyA ? tA : -0.0f;
      nz.y = noisyB ? tB : -0.0f;
      if (live) {
        gA[0] = gA[1]; gA[1] = gA[2]; gA[2] = gA[3];
        gB[0] = gB[1]; gB[1] = gB[2]; gB[2] = gB[3];
        nctr += 1u;
      }
    }
    return nz;
  };
  // one step (k_ring_pair's `advance`): `have_act` false = rl_actions None (the warm-up steps of a reset)
  auto advance = [&](bool live, bool have_act, float a_rl) {
    f2 acc = idm_pair<FAST, FAST>(v, vl, h, p, two_sqrt_ab, rc_ab, rc_v0, one);
    if constexpr (NOISE) acc = pk_add(acc, noise_term(live));
    const float a_cl = clip(a_rl);
    acc.x = rlA ? a_cl : acc.x;
    acc.y = rlB ? a_cl : acc.y;
    const bool cmdA = !rlA || have_act, cmdB = !rlB || have_act;
    const f2 acc_s = sumo_acc_pair<FAST>(v, vl, h, sc, one);
    if (MIXED) {
      const double aA = double(acc.x), aB = double(acc.y);
      const double nA = tmax(vdA + aA * dt64, 0.0), nB = tmax(vdB + aB * dt64, 0.0);
      double cA = vdA + (nA - vdA) * ramp64, cB = vdB + (nB - vdB) * ramp64;
      const double vsA = vdA + double(acc_s.x) * dt64, vsB = vdB + double(acc_s.y) * dt64;
      cA = tmin(cA, tmax(vsA, floor0A)); cB = tmin(cB, tmax(vsB, floor0B));
      cA = tmin(cA, vdA + adtA); cB = tmin(cB, vdB + adtB);
      cA = tmax(cA, vdA - ddtA); cB = tmax(cB, vdB - ddtB);
      cA = cmdA ? cA : tmax(vsA, 0.0);
      cB = cmdB ? cB : tmax(vsB, 0.0);
      const double xA = xdA + cA * dt64, xB = xdB + cB * dt64;
      const double wA = xA >= L64 ? xA - L64 : xA, wB = xB >= L64 ? xB - L64 : xB;
      if (live) {
        vdA = cA; vdB = cB;
        xdA = wA; xdB = wB;
      }
      v = f2{float(vdA), float(vdB)};
    } else {
      f2 nv = pk_add(v, pk_mul(acc, dt2));
      nv.x = hmax(nv.x, 0.0f);
      nv.y = hmax(nv.y, 0.0f);
      f2 vc = pk_add(v, pk_mul(pk_sub(nv, v), ramp2));
      const f2 vs = pk_add(v, pk_mul(acc_s, dt2));
      const f2 cap1 = pk_add(v, sc.adt), flo = pk_sub(v, sc.ddt);
      vc.x = hmin(vc.x, hmax(sc.floor0.x, vs.x));
      vc.y = hmin(vc.y, hmax(sc.floor0.y, vs.y));
      vc.x = hmax(hmin(vc.x, cap1.x), flo.x);
      vc.y = hmax(hmin(vc.y, cap1.y), flo.y);
      const float zA = hmax(0.0f, vs.x), zB = hmax(0.0f, vs.y);
      vc.x = cmdA ? vc.x : zA;
      vc.y = cmdB ? vc.y : zB;
      const f2 xn = pk_add(x, pk_mul(vc, dt2));
      const f2 xw = pk_sub(xn, L2);
      f2 xq;
      xq.x = nonneg_else(xw.x, xn.x);
      xq.y = nonneg_else(xw.y, xn.y);
      if (live) {
        v = vc;
        x = xq;
      }
    }
    if (live) tcount += 1;
    h = headway();
    vl = f2{v.y, next_a<ROW>(v.x, last, lane)};
  };

  // WaveAttenuationPOEnv.get_state of the current snapshot (k_ring_pair's write_obs): computed by the RL vehicle's
  // lane, handed to the row through LDS (the policy's input), stored by that lane
  const bool poA = valid && rlA, poB = valid && rlB;
  const double rc15 = 1.0 / 15.0, pml64 = double(s.po_max_length), rc_pml64 = 1.0 / pml64;
  float o0 = 0.0f, o1 = 0.0f, o2 = 0.0f;
  // float32: the three quotients are exact divisions through float64 (five instructions each, issued for one lane):
  // the RL vehicle's lane hands its second and third numerator to its two neighbours (row rotations) and the three lanes
  // divide one value each -- one division sequence per step instead of three
  const unsigned long long po_m = __ballot(poA || poB);
  const int k_po = po_m ? (__builtin_ctzll(po_m) & (ROW - 1)) : 0;
  const int po_c = (k - k_po) & (ROW - 1);                               // 0: the RL vehicle's lane, 1 / 2: its helpers
  const double po_div = po_c == 2 ? pml64 : 15.0, po_rc = po_c == 2 ? rc_pml64 : rc15;
  // (hand-off to the row through the LDS crossbar, ds_bpermute: one round trip and no memory, where the first version
  // wrote the three values to LDS and read them back between two wave barriers)
  const int src0 = (lane - k) + k_po, src1 = (lane - k) + ((k_po + 1) & (ROW - 1)), src2 = (lane - k) + ((k_po + 2) & (ROW - 1));
  auto observe = [&](float* orow) {
    if constexpr (!MIXED && ROW == 16) {
      const float v_me = poB ? v.y : v.x, v_ld = poB ? vl.y : vl.x, d_me = poB ? dgap.y : dgap.x;
      const float n1 = dpp<0x120 + 1>(v_ld - v_me), n2 = dpp<0x120 + 2>(d_me);      // row_ror: lane i <- lane i - 1 / i - 2
      const float n = po_c == 0 ? v_me : (po_c == 1 ? n1 : n2);
      const float q = div_via_f64(n, po_div, po_rc);
      if (rvalid && po_c < 3) orow[po_c] = q;
      row_bcast3(k_po, q, q, q, o0, o1, o2);
    } else {
      float q0, q1, q2;
      if (MIXED) {
        const double vdn = next_a<ROW>(vdA, last, lane);
        const double v_me = poB ? vdB : vdA, v_ld = poB ? vdn : vdB, d_me = poB ? dgB : dgA;
        q0 = float(v_me * rc15);
        q1 = float((v_ld - v_me) * rc15);
        q2 = float(d_me * rc_pml64);
      } else {
        const float v_me = poB ? v.y : v.x, v_ld = poB ? vl.y : vl.x, d_me = poB ? dgap.y : dgap.x;
        q0 = div_via_f64(v_me, 15.0, rc15);
        q1 = div_via_f64(v_ld - v_me, 15.0, rc15);
        q2 = div_via_f64(d_me, pml64, rc_pml64);
      }
      if (poA || poB) {
        orow[0] = q0;
        orow[1] = q1;
        orow[2] = q2;
      }
      o0 = __shfl(q0, src0, 64);
      o1 = __shfl(q1, src0, 64);
      o2 = __shfl(q2, src0, 64);
    }
  };

  const size_t R = size_t(s.R);
  NoiseBlock<float> act_draws;
  act_draws.init();
  observe(obs + size_t(rr) * 3);
  for (int step = 0; step < num_steps; ++step) {
    // ---- policy -> action --------------------------------------------------------------------------------------
    float mu, ls, a, lp;
    policy_eval<ROW>(pv, &PL, k, o0, o1, o2, mu, ls);
    policy_sample(pv, s.rep0 + uint32_t(rr), pctr, mu, ls, a, lp, &act_draws);
    pctr += 1u;
    if (rvalid && k == 0) {
      act[size_t(step) * R + rr] = a;
      logp[size_t(step) * R + rr] = lp;
    }
    // ---- Env.step ------------------------------------------------------------------------------------------------
    advance(true, true, a);
    const f2 hc = pk_sub(h, gap2), vb = pk_sub(v, f2{-100.0f, -100.0f});        // sign masks (k_ring_pair's terms)
    const unsigned fl = ((__builtin_bit_cast(unsigned, hmin(hc.x, hc.y)) >> 31) |
                         ((__builtin_bit_cast(unsigned, hmin(vb.x, vb.y)) >> 31) << 1)) & (valid ? 3u : 0u);
    const unsigned fany = seg_or<ROW>(fl);
    const bool crashed = (fany & 1u) != 0u;
    const bool bad = (fany & 2u) != 0u || crashed;
    const float sv = seg_sum<ROW>(valid ? v.x + v.y : 0.0f);
    const float mean_v = div_via_f64(sv, double(N), 1.0 / double(N));
    const float mean_a = tabs(clip(a));                     // (one column: the sum is the value, / 1)
    float reward = div_via_f64(4.0f * mean_v, 20.0, 1.0 / 20.0);
    if (mean_a > 0.0f) reward = reward + 4.0f * (0.0f - mean_a);
    reward = bad ? 0.0f : reward;
    const uint8_t dflag = done_flag(tcount >= s.step_limit, crashed);
    if (rvalid && k == 0) {
      rew[size_t(step) * R + rr] = reward;
      done[size_t(step) * R + rr] = dflag;
    }
    // ---- Env.reset of a finished episode (what VecFlowEnv.capture(reset_done=True) does with a masked fs_reset_dev):
    // placement, the pending ring length, warm-up steps with rl_actions = None; the other replicas of the wave wait
    const bool fin = reset_done && dflag != 0;
    if (__ballot(fin) != 0ull) {
      if (fin) {
        load_state(s.init_pos, s.init_vel);
        L = s.init_ring_len[rr] + T(4) * s.jlen;
        L2 = splat(float(L));
        L64 = double(L);
        tcount = 0;
      }
      h = headway();
      vl = f2{v.y, next_a<ROW>(v.x, last, lane)};
#pragma unroll 1
      for (int w = 0; w < warmup_steps; ++w) advance(fin, false, 0.0f);
      if (fin && valid && kk == 0) const_cast<T*>(s.ring_len)[rr] = s.init_ring_len[rr];
    }
    observe(obs + (size_t(step + 1) * R + rr) * 3);
  }

  if (valid) {
    if (MIXED) {
      s.pos[idx] = T(xdA); s.pos[idx + 1] = T(xdB);
      s.vel[idx] = T(vdA); s.vel[idx + 1] = T(vdB);
    } else {
      s.pos[idx] = T(x.x); s.pos[idx + 1] = T(x.y);
      s.vel[idx] = T(v.x); s.vel[idx + 1] = T(v.y);
    }
    if (kk == 0) {
      s.time[rr] = tcount;
      pv.ctr[rr] = pctr;
      if (NOISE) s.noise_ctr[rr] = nctr;
    }
  }
}


// K x (policy -> action -> Env.step [-> reset of a finished episode]) on a segment-table loop (the figure eight: BASELINE's
// C3, examples/exp_configs/rl/singleagent/singleagent_figure_eight.py) with ONE RL vehicle.  HEAD 1: WaveAttenuationPOEnv
// (observation 3: BASELINE's pairing), HEAD 0: AccelEnv (observation 2 N: the reference's own pairing -- every lane feeds its
// vehicle's speed and position into the first layer, policy_eval<., WIDE>).  The simulator part is k_rollout_loop's step
// (flowsim_fig8.h), statement by statement: the same model functions, crossing rule, segment cursor, flag word, observation
// quotients and reward expressions, so a fragment equals eager stepping (fs_policy_act_dev, fs_step_dev -- which runs
// k_rollout_loop --, masked fs_reset_dev) bit for bit (tests/test_policy_gpu.py).  Resets inside the fragment: placement only
// (warmup_steps = 0: Sim::launch_policy refuses anything else).
template <int HEAD, bool DELTA4, bool FASTC>
__global__ __launch_bounds__(256) void k_loop_policy(DevView<float> s, PolicyView pv, int num_steps, int reset_done,
                                                     float* __restrict__ obs, float* __restrict__ act,
                                                     float* __restrict__ logp, float* __restrict__ rew,
                                                     uint8_t* __restrict__ done) {
  typedef float T;
  typedef float X;
  constexpr int SEG = 16, RPW = 4;
  __shared__ X tab_start[FS_MAX_SEGMENTS + 2], tab_fs[FS_MAX_SEGMENTS + 2], tab_sl[FS_MAX_SEGMENTS + 2];
  __shared__ PolicyLds PL;
  policy_load(pv, &PL, threadIdx.x, blockDim.x);
  const int lane = threadIdx.x & 63;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int seg = lane / SEG;
  const int i = lane % SEG;
  const int r = wave * RPW + seg;
  const int N = s.N;
  const bool rvalid = r < s.R;
  const bool valid = rvalid && i < N;
  const int rr = rvalid ? r : s.R - 1;
  const int ii = i < N ? i : N - 1;
  const size_t idx = size_t(rr) * N + ii;
  const bool wrap_lead = (i + 1 >= N);
  constexpr bool has = true;                          // N > 1 (host-checked)
  const int flags = s.flags;

  if (threadIdx.x < FS_MAX_SEGMENTS + 2) {
    const int q = threadIdx.x, qq = q < FS_MAX_SEGMENTS ? q : 0;
    tab_start[q] = q < s.nseg ? s.seg_start[qq] : X(3.0e38);
    tab_fs[q] = q < s.nseg ? s.seg_flow_start[qq] : X(0);
    tab_sl[q] = q < s.nseg ? s.seg_flow_slope[qq] : X(0);
  }
  __syncthreads();

  Slot<T> sl;
  sl.ctrl = s.ctrl[ii];
  sl.failsafe = 0;
  sl.speed_mode = s.speed_mode[ii];
  sl.rl_index = s.rl_index[ii];
  sl.pis_index = -1;
#pragma unroll
  for (int k = 0; k < FS_MAX_CTRL_PARAMS; ++k) sl.p[k] = T(s.p[k * N + ii]);
  sl.noise = T(s.noise[ii]);
  sl.delay = T(0);
  sl.max_accel = T(s.max_accel[ii]);
  sl.max_decel = T(s.max_decel[ii]);
  sl.length = T(s.length[ii]);
  sl.sumo_tau = T(s.sumo_tau[ii]);
  sl.sumo_min_gap = T(s.sumo_min_gap[ii]);
  sl.sumo_max_speed = T(s.sumo_max_speed[ii]);
  const X len_me = s.length[ii];
  const X len_lead = lead16(len_me, wrap_lead);

  const X L = s.ring_len[rr] + X(4) * s.jlen;
  int tcount = s.time[rr];
  const bool any_noise = (flags & FLAG_HAS_NOISE) != 0;
  uint32_t nctr = any_noise ? s.noise_ctr[rr] : 0u;
  uint32_t pctr = pv.ctr[rr];
  const bool noisy = any_noise && sl.noise > T(0) && sl.ctrl != FS_CTRL_RL && sl.ctrl != FS_CTRL_SIM;

  X x = s.pos[idx];
  X v = s.vel[idx];
  int k = 0;
  X c_st, c_next, c_fs, c_sl;
  auto cursor = [&]() {                                    // the segment of x and its table row
    k = 0;
    for (int q = 1; q < s.nseg; ++q) k = (x >= tab_start[q]) ? q : k;
    c_st = tab_start[k]; c_next = tab_start[k + 1]; c_fs = tab_fs[k]; c_sl = tab_sl[k];
  };
  cursor();
  X xl, vl, d;
  T h;
  const X Lv = in_vgpr(L);
  auto snapshot = [&]() {
    xl = lead16(x, wrap_lead);
    vl = lead16(v, wrap_lead);
    d = wrap_up(xl - x, Lv);
    h = has ? T(d - len_lead) : T(1000);
  };
  snapshot();

  const X dt = in_vgpr(X(s.dt)), ramp = in_vgpr(X(s.ramp));
  const T crash_gap = in_vgpr(T(s.crash_gap)), target_v = in_vgpr(T(s.target_velocity));
  const X ja_in = in_vgpr(X(s.ja_in)), ja_out = in_vgpr(X(s.ja_out)), jb_in = in_vgpr(X(s.jb_in)), jb_out = in_vgpr(X(s.jb_out));
  const X look = in_vgpr(X(s.j_lookahead)), tgap = in_vgpr(X(s.j_time_gap));
  const X za_lo = in_vgpr(X(s.za_lo)), za_hi = in_vgpr(X(s.za_hi)), zb_lo = in_vgpr(X(s.zb_lo)), zb_hi = in_vgpr(X(s.zb_hi));
  const T max_cost = in_vgpr(T(s.max_cost));
  const DivC d_ms = make_divc(T(s.max_speed)), d_L = make_divc(T(L)), d_15 = make_divc(15.0f), d_po = make_divc(T(s.po_max_length));
  // the reward's quotients (k_rollout_loop divides in float32 once per block of four steps; here every step does, so the
  // constant divisors take the float64 route: the correctly rounded quotient either way, half the instructions)
  const DivC d_N = make_divc(T(s.N)), d_nrl = make_divc(T(s.num_rl > 0 ? s.num_rl : 1)), d_20 = make_divc(20.0f),
             d_mc = make_divc(T(s.max_cost) + T(1.1920928955078125e-07));
  IdmC ic;
  ic.p1 = sl.p[1]; ic.p2 = sl.p[2]; ic.p4 = sl.p[4]; ic.p5 = sl.p[5];
  ic.v0 = make_divc(sl.p[0]);
  ic.two_sqrt = make_divc(T(2) * tsqrt(sl.p[2] * sl.p[3]));
  SumoC sc;
  sc.min_gap = sl.sumo_min_gap; sc.tau = sl.sumo_tau; sc.max_accel = sl.max_accel;
  sc.two_sqrt = make_divc(T(2) * tsqrt(sl.max_accel * sl.max_decel));
  sc.max_speed = make_divc(sl.sumo_max_speed);
  const unsigned seg_internal = s.seg_internal;
  const bool junction_on = s.junction_on != 0, need_sumo = (flags & FLAG_NEED_SUMO) != 0;
  const bool gated = s.junction_mode && sl.ctrl != FS_CTRL_RL && sl.ctrl != FS_CTRL_SIM;
  const T clip_lo = in_vgpr(s.clip_actions != 0 ? T(s.act_lo) : T(-3.0e38)), clip_hi = in_vgpr(s.clip_actions != 0 ? T(s.act_hi) : T(3.0e38));
  const bool rl_lane = sl.ctrl == FS_CTRL_RL, sim_lane = sl.ctrl == FS_CTRL_SIM;
  const int num_rl = s.num_rl;
  const bool obs_lane = HEAD == 1 ? (valid && rl_lane && sl.rl_index == 0) : valid;
  const unsigned valid_bits = valid ? 0x3Fu : 0u;
  const unsigned gate_u = gated ? 1u : 0u, cmd_rl = rl_lane ? 1u : 0u, cmd_other = (!rl_lane && !sim_lane) ? 1u : 0u,
                 sm1_u = unsigned(sl.speed_mode) & 1u;
  const bool sm1_lane = (sl.speed_mode & 1) != 0;
  const X adt_c = (sl.speed_mode & 2) ? X(s.max_accel[ii]) * dt : X(3.0e38), ddt_c = (sl.speed_mode & 4) ? X(s.max_decel[ii]) * dt : X(3.0e38);
  T g4[4] = {T(0), T(0), T(0), T(0)};
  if (any_noise && (nctr & 3u) != 0u && noisy) {
    gauss4<T>(s.seed_lo, s.seed_hi, s.rep0 + uint32_t(rr), uint32_t(ii), nctr >> 2, g4, s.noise_exact != 0);
    for (uint32_t q = 0; q < (nctr & 3u); ++q) { g4[0] = g4[1]; g4[1] = g4[2]; g4[2] = g4[3]; }
  }
  X prev_v = v;
  T last_acc = T(0);
  auto junction_flags = [&](X xx, X vv) -> unsigned {
    const unsigned busy_a = sm_in(xx, ja_in - tgap * vv, ja_out + len_me);
    const unsigned in_b = sm_in(xx, jb_in, jb_out + len_me);
    return (busy_a >> 31) | ((in_b >> 31) << 1);
  };
  unsigned jf = junction_on ? seg_or<SEG>(junction_flags(x, v) & valid_bits) : 0u;

  // the observation of the current state: stored, and handed to the row as the policy's input.  PO head: the RL vehicle's
  // three values reach every lane of its row; AccelEnv head: every lane keeps its own two (inputs ii and N + ii)
  const int obs_dim = HEAD == 1 ? 3 : 2 * N;
  const unsigned long long rl_m = __ballot(valid && rl_lane);
  const int k_rl = rl_m ? (__builtin_ctzll(rl_m) & (SEG - 1)) : 0;
  const int src_rl = (lane - i) + k_rl;
  float o0 = 0.0f, o1 = 0.0f, o2 = 0.0f;
  auto observe = [&](float* orow) {
    if (HEAD == 1) {
      const float po0 = divc(v, d_15), po1 = divc(vl - v, d_15), po2 = divc(d, d_po);      // wave_attenuation.py:248-269
      if (obs_lane) { orow[0] = po0; orow[1] = po1; orow[2] = po2; }
      row_bcast1x3(k_rl, po0, po1, po2, o0, o1, o2);
    } else {
      const X xo = c_fs + c_sl * (x - c_st);
      const float po0 = divc(v, d_ms), po1 = divc(xo, d_L);                                   // accel.py:116-123
      if (obs_lane) { orow[ii] = po0; orow[N + ii] = po1; }
      o0 = valid ? po0 : 0.0f;
      o1 = valid ? po1 : 0.0f;
    }
  };

  const size_t R = size_t(s.R);
  NoiseBlock<float> act_draws;
  act_draws.init();
  observe(obs + size_t(rr) * obs_dim);
  for (int step = 0; step < num_steps; ++step) {
    // ---- policy -> action --------------------------------------------------------------------------------------
    float mu, ls, a, lp;
    if (HEAD == 1) policy_eval<SEG>(pv, &PL, i, o0, o1, o2, mu, ls);
    else policy_eval<SEG, true>(pv, &PL, i, o0, o1, 0.0f, mu, ls);
    policy_sample(pv, s.rep0 + uint32_t(rr), pctr, mu, ls, a, lp, &act_draws);
    pctr += 1u;
    if (rvalid && i == 0) {
      act[size_t(step) * R + rr] = a;
      logp[size_t(step) * R + rr] = lp;
    }
    // ---- Env.step: k_rollout_loop's step ---------------------------------------------------------------------------
    bool on_a = false, on_b = false, on_any = false, on_both = false;
    if (junction_on) {
      const unsigned on_b_m = sm_in(x, jb_in - look, jb_in) & (jf << 31);
      const unsigned on_a_m = sm_in(x, ja_in - look, ja_in) & (jf << 30);
      on_b = sm_true(on_b_m);
      on_a = sm_true(on_a_m);
      on_any = sm_true(on_a_m | on_b_m);
      on_both = sm_true(on_a_m & on_b_m);
    }
    const unsigned on_edge_u = 1u ^ (gate_u & (seg_internal >> k));
    const unsigned commanded_u = (on_edge_u & cmd_other) | cmd_rl;
    const bool commanded = commanded_u != 0u;
    T acc;
    {
      if (any_noise) {
        if (__ballot(noisy && (nctr & 3u) == 0u) != 0ull) {
          if (noisy && (nctr & 3u) == 0u)
            gauss4<T>(s.seed_lo, s.seed_hi, s.rep0 + uint32_t(rr), uint32_t(ii), nctr >> 2, g4, s.noise_exact != 0);
        }
      }
      T ai = idm_fast<DELTA4, FASTC>(T(v), T(vl), h, has, ic);
      if (any_noise) {
        const T an = ai + sl.noise * g4[0];
        ai = noisy ? an : ai;
        g4[0] = g4[1]; g4[1] = g4[2]; g4[2] = g4[3];
      }
      const T arl = hmin(hmax(T(a), clip_lo), clip_hi);
      acc = rl_lane ? arl : (sim_lane ? T(0) : ai);
    }
    X next_vel = xmax(v + X(acc) * dt, X(0));
    X vc = v + (next_vel - v) * ramp;
    X v_new = vc;
    if (need_sumo) {
      X v_sumo = sumo_fast<FASTC>(v, vl, h, has, dt, sc);
      vc = xmin(vc, sm1_lane ? v_sumo : X(3.0e38));
      vc = xmin(vc, v + adt_c);
      vc = xmax(vc, v - ddt_c);
      v_new = commanded ? vc : v_sumo;
    }
    if (junction_on) {
      if (__ballot(on_any) != 0ull) {
        const X line = on_b ? jb_in - x : ja_in - x;
        X cap = sumo_fast<FASTC>(v, X(0), T(line), true, dt, sc);
        cap = on_any ? cap : X(3.0e38);
        if (__ballot(on_both) != 0ull) {
          const X cap_a = sumo_fast<FASTC>(v, X(0), T(ja_in - x), true, dt, sc);
          cap = xmin(cap, on_both ? cap_a : X(3.0e38));
        }
        const bool cap_applies = (sm1_u | (commanded_u ^ 1u)) != 0u;
        v_new = xmin(v_new, cap_applies ? cap : X(3.0e38));
      }
    }
    X x_new = x + v_new * dt;
    x_new = wrap_down(x_new, Lv);
    prev_v = v;
    last_acc = acc;
    x = x_new;
    v = v_new;
    tcount += 1;
    nctr += 1u;
    // the segment cursor (k_rollout_loop advances it by compares; the position decides the segment either way)
    if (__ballot((x < c_st) || (x >= c_next)) != 0ull) cursor();
    snapshot();
    unsigned f2_ = sm_lt(h, crash_gap) >> 31;
    if (junction_on) {
      f2_ |= (sm_in(x, za_lo, za_hi) >> 31) << 1;
      f2_ |= (sm_in(x, zb_lo, zb_hi) >> 31) << 2;
    }
    f2_ |= (sm_lt(v, X(-100)) >> 31) << 3;
    if (junction_on) f2_ |= junction_flags(x, v) << 4;
    f2_ &= valid_bits;
    f2_ = seg_or<SEG>(f2_);
    jf = (f2_ >> 4) & 3u;
    const bool crashed = ((f2_ | ((f2_ >> 1) & (f2_ >> 2))) & 1u) != 0u;
    const bool bad = (((f2_ >> 3) & 1u) != 0u) || crashed;
    // ---- reward (the block form's transposed_sum is seg_sum's tree) --------------------------------------------------
    T reward;
    if (HEAD == 1) {                                               // wave_attenuation.py:113-139
      const T racc = seg_sum<SEG>(valid ? T(v) : T(0));
      // (k_rollout_loop sums |clip(a)| over the lanes of the RL columns: one column, one non-zero term -- the term itself)
      const T racc2 = tabs(hmin(hmax(T(a), clip_lo), clip_hi));
      const T mean_v = divc(racc, d_N);
      const T mean_a = divc(racc2, d_nrl);
      reward = divc(T(4.0) * mean_v, d_20);
      if (mean_a > T(0)) reward = reward + T(4) * (T(0) - mean_a);
      reward = bad ? T(0) : reward;
    } else {                                                       // rewards.py:6-59
      const T dv = valid ? T(v) - target_v : T(0);
      const T cost = tsqrt(seg_sum<SEG>(dv * dv));
      reward = divc(tmax(max_cost - cost, T(0)), d_mc);
      reward = bad ? T(0) : reward;
    }
    const uint8_t dflag = done_flag(tcount >= s.step_limit, crashed);
    if (rvalid && i == 0) {
      rew[size_t(step) * R + rr] = reward;
      done[size_t(step) * R + rr] = dflag;
    }
    // ---- Env.reset of a finished episode (masked fs_reset_dev: the placement; the noise stream runs on) ---------------
    const bool fin = reset_done && dflag != 0;
    if (__ballot(fin) != 0ull) {
      if (fin) {
        x = s.init_pos[idx];
        v = s.init_vel[idx];
        tcount = 0;
      }
      cursor();
      snapshot();
      jf = junction_on ? seg_or<SEG>(junction_flags(x, v) & valid_bits) : 0u;
    }
    observe(obs + (size_t(step + 1) * R + rr) * obs_dim);
  }

  if (valid) {
    s.pos[idx] = x;
    s.vel[idx] = v;
    if (s.track_aux && num_steps > 0) {
      s.prev_vel[idx] = prev_v;
      s.accel[idx] = last_acc;
    }
    if (ii == 0) {
      s.time[rr] = tcount;
      pv.ctr[rr] = pctr;
      if (any_noise) s.noise_ctr[rr] = nctr;
    }
  }
}

}  // namespace fs
