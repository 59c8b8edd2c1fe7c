// flowsim_launch.h -- Sim<T>::launch_seg<SEG> / launch_wide<W>: which step kernel a launch takes.  Included by
// flowsim_part.hip only, so that each (precision, SEG) pair and its kernels compile in an object of their own.
#pragma once
#include "flowsim_sim.h"
#ifdef FS_PART_QUEUE
#include "flowsim_queue.h"
#include "flowsim_dropq.h"
#endif

namespace fsim {

#ifdef FS_PART_QUEUE
  template <typename T>
  int Sim<T>::launch_queue(int num_steps, const float* actions, size_t act_stride, float* obs, float* rew, uint8_t* done,
                           int obs_every_step) {
    if constexpr (std::is_same<T, float>::value) {
      last_kernel = "k_merge_queue";
      const bool noise = (dv.flags & fs::FLAG_HAS_NOISE) != 0, act = actions != nullptr && ov.ma_apply_actions != 0;
#define FS_QUEUE(NZ_, ACT_)                                                                                        \
  hipLaunchKernelGGL((fs::k_merge_queue<NZ_, ACT_>), dim3(dv.R), dim3(64), 0, stream, dv, ov, qc, num_steps, actions, \
                     act_stride, obs, rew, done, obs_every_step)
      if (noise) { if (act) FS_QUEUE(true, true); else FS_QUEUE(true, false); }
      else { if (act) FS_QUEUE(false, true); else FS_QUEUE(false, false); }
#undef FS_QUEUE
      HIP_TRY(hipGetLastError());
      return FS_OK;
    } else {
      return fail(FS_ERR_UNSUPPORTED, "k_merge_queue is a float32 kernel");
    }
  }
  template <typename T>
  int Sim<T>::launch_dropq(int num_steps, const float* actions, size_t act_stride, float* obs, float* rew, uint8_t* done,
                           int obs_every_step) {
    if constexpr (std::is_same<T, float>::value) {
      last_kernel = "k_drop_queue";
      qflag_armed = true;
      if (dv.env == FS_ENV_BOTTLENECK_DV)
        hipLaunchKernelGGL((fs::k_drop_queue<true>), dim3(dv.R), dim3(256), 0, stream, dv, ov, qc, d_qflag, num_steps, actions,
                           act_stride, obs, rew, done, obs_every_step);
      else
        hipLaunchKernelGGL((fs::k_drop_queue<false>), dim3(dv.R), dim3(256), 0, stream, dv, ov, qc, d_qflag, num_steps, actions,
                           act_stride, obs, rew, done, obs_every_step);
      HIP_TRY(hipGetLastError());
      return FS_OK;
    } else {
      return fail(FS_ERR_UNSUPPORTED, "k_drop_queue is a float32 kernel");
    }
  }
#endif

  // more than 64 slots per replica (lane-drop network): one workgroup of W waves per replica
  template <typename T>
  template <int W>
  int Sim<T>::launch_wide(int num_steps, const uint8_t* mask, const float* actions, size_t act_stride, float* obs,
                  float* rew, uint8_t* done, int obs_every_step) {
    // float32 exists twice (CSET = 1: IDM / RL / Sim slots only); num_paths = 8 is the scaling-2 network
    constexpr int C1 = std::is_same<T, float>::value ? 1 : 0;
    if constexpr (std::is_same<T, float>::value) {
      if (dropq_ok(mask, num_steps)) return launch_dropq(num_steps, actions, act_stride, obs, rew, done, obs_every_step);
    }
    const bool cset = C1 == 1 && (dv.flags & fs::FLAG_IDM_SET) && !force_generic && open_div_ok;
#define FS_WIDE(P_, C_)                                                                                          \
  hipLaunchKernelGGL((fs::k_steps_wide<T, W, C_, P_>), dim3(dv.R), dim3(64 * W), 0, stream, dv, ov, num_steps, mask, \
                     actions, act_stride, obs, rew, done, obs_every_step, after_reset)
    last_kernel = "k_steps_wide";
    if (cfg.num_paths == 8) { if (cset) FS_WIDE(8, C1); else FS_WIDE(8, 0); }
    else { if (cset) FS_WIDE(4, C1); else FS_WIDE(4, 0); }
#undef FS_WIDE
    HIP_TRY(hipGetLastError());
    return FS_OK;
  }

  template <typename T>
  template <int SEG>
  int Sim<T>::launch_seg(int num_steps, const uint8_t* mask, const float* actions, size_t act_stride, float* obs,
                 float* rew, uint8_t* done, int obs_every_step) {
    constexpr int RPW = 64 / SEG;
    const int blocks = (dv.R + RPW - 1) / RPW;
    if (has_user_ctrl && !fs::kHasUserController)
      return fail(FS_ERR_UNSUPPORTED, "FS_CTRL_USER: this library was built without a user controller "
                                      "(flow_amd.build.build_user / flow_amd.controllers.CompiledController)");
    if (open_net) {
      if constexpr (std::is_same<T, float>::value) {
        if (queue_ok(mask, num_steps)) return launch_queue(num_steps, actions, act_stride, obs, rew, done, obs_every_step);
        if (dropq_ok(mask, num_steps)) return launch_dropq(num_steps, actions, act_stride, obs, rew, done, obs_every_step);
      }
      // the float32 instantiations exist twice: CSET = 1 for populations of IDM / RL / Sim slots only
      const bool cset = std::is_same<T, float>::value && (dv.flags & fs::FLAG_IDM_SET) && !force_generic && open_div_ok;
      // FS_MIXED: the float64 kernel with the float32 car-following models (CSET = 2); a population outside their
      // premises steps in plain float64
      const bool mset = std::is_same<T, double>::value && mixed && (dv.flags & fs::FLAG_IDM_SET) && !force_generic && open_div_ok;
      constexpr int C2 = std::is_same<T, double>::value ? 2 : 0;
      // plain float64 with an IDM / RL / Sim population: the branch-free controller selection (CSET = 1) in float64
      // arithmetic -- the same operations as the generic instantiation, without its per-controller exec-mask branches
      const bool dset = std::is_same<T, double>::value && !mixed && (dv.flags & fs::FLAG_IDM_SET) && !force_generic;
#define FS_OPEN__(P_, C_, PR_, PO_)                                                                              \
  hipLaunchKernelGGL((fs::k_steps_open<T, SEG, P_, C_, PR_, PO_>), dim3(blocks), dim3(64), 0, stream, dv, ov,        \
                     num_steps, mask, actions, act_stride, obs, rew, done, obs_every_step, after_reset)
#define FS_OPEN_(P_, C_, PR_) do { if (P_ == 2 && dv.env == FS_ENV_MERGE_PO) FS_OPEN__(P_, C_, PR_, (P_ == 2));    \
                                   else FS_OPEN__(P_, C_, PR_, false); } while (0)
#define FS_OPEN(P_, C_) do { if (ov.n_prob > 0) FS_OPEN_(P_, C_, true); else FS_OPEN_(P_, C_, false); } while (0)
      last_kernel = "k_steps_open";
      if (cfg.network == FS_NET_BOTTLENECK) {
        // the lane-drop heads need more than 32 slots (fs_create checks it): only the 64-lane segment is built
        if constexpr (SEG == 64) {
          if (cset || dset) FS_OPEN(4, 1); else if (mset) FS_OPEN(4, C2); else FS_OPEN(4, 0);
        } else {
          return fail(FS_ERR_UNSUPPORTED, "fs_step: FS_NET_BOTTLENECK runs on 64-lane segments only");
        }
      } else {
        if (cset || dset) FS_OPEN(2, 1); else if (mset) FS_OPEN(2, C2); else FS_OPEN(2, 0);
        if (mset) last_kernel = "k_steps_open<mixed>";
      }
#undef FS_OPEN
#undef FS_OPEN_
#undef FS_OPEN__
      HIP_TRY(hipGetLastError());
      return FS_OK;
    }
    if (dv.num_lanes > 1 || dv.env == FS_ENV_LANE_CHANGE_ACCEL || dv.env == FS_ENV_LANE_CHANGE_ACCEL_PO) {
      last_kernel = "k_steps_ml";
      const bool lcpo = dv.env == FS_ENV_LANE_CHANGE_ACCEL_PO;
#define FS_ML(LC_, PO_) hipLaunchKernelGGL((fs::k_steps_ml<T, SEG, LC_, PO_>), dim3(blocks), dim3(64), 0, stream, dv, num_steps, \
                                           mask, actions, act_stride, obs, rew, done, obs_every_step)
      if (dv.lc_enabled) { if (lcpo) FS_ML(true, true); else FS_ML(true, false); }
      else { if (lcpo) FS_ML(false, true); else FS_ML(false, false); }
#undef FS_ML
      HIP_TRY(hipGetLastError());
      return FS_OK;
    }
    // two vehicles per lane (flowsim_pair.h): even N; with k_ring_pair the stepping kernels of a FS_MIXED ring handle
    constexpr int ROW = SEG >= 16 ? SEG / 2 : 8;
    const bool pair_noise = std::is_same<T, float>::value && !mixed;      // the noisy form exists in float32 only
    const bool pair_ok = fast_ok(mask, num_steps, true, pair_noise) && (obs_every_step || num_steps == 1) && dv.N >= 2 &&
                         (dv.N % 2) == 0 && actions == nullptr && !no_pair &&
                         size_t(dv.R) * 2 * dv.N * sizeof(float) * 16 < (size_t(1) << 32);   // 32-bit offsets in a block
    // closed loops with a segment table (figure eight): the rollout kernel of flowsim_fig8.h
    // (FS_MIXED handles: its float64-state instantiation; whatever that does not cover -- resets, masks, warm-up, single
    // vehicles -- steps on the generic float64 kernel, which is the reference's arithmetic)
    if constexpr (SEG == 16) {
      constexpr bool MX = !std::is_same<T, float>::value;
      const int f = dv.flags;
      if (!MX || mixed) {
      const bool ma_loop = !MX && dv.env == FS_ENV_ACCEL_PO_MA;        // MultiAgentAccelPOEnv (multiagent_figure_eight.py)
      const bool head_ok = (dv.env == FS_ENV_ACCEL && !dv.evaluate) || dv.env == FS_ENV_WAVE_ATTENUATION_PO || ma_loop;
      if (dv.nseg > 0 && (f & fs::FLAG_IDM_SET) && !(f & fs::FLAG_HAS_FAILSAFE) && head_ok &&
          dv.integrator == FS_EULER && dv.sims_per_step == 1 && mask == nullptr &&
          !dv.sort_vehicles && dv.obs_perm == nullptr && num_steps > 0 && (obs_every_step || num_steps == 1) &&
          dv.N > 1 && loop_div_ok && !force_generic && !no_loop_kernel) {
        const int waves = (dv.R + 3) / 4;
        const dim3 grid((waves + 3) / 4), block(256);
        last_kernel = "k_rollout_loop";
#define FS_LOOP(H_, D_)                                                                                       \
  hipLaunchKernelGGL((fs::k_rollout_loop<H_, D_, false, MX>), grid, block, 0, stream, dv, num_steps, actions, act_stride, \
                     obs, rew, done)
        const bool full = (f & fs::FLAG_HAS_NOISE) && (f & fs::FLAG_NEED_SUMO) && dv.junction_on && actions != nullptr &&
                          loop_delta4 && !no_loop_full && loop_fastc_ok();
        if (ma_loop) {
          if constexpr (!MX) {
            last_kernel = full ? "k_rollout_loop<FULL,AccelMA>" : "k_rollout_loop<AccelMA>";
            if (full) hipLaunchKernelGGL((fs::k_rollout_loop<2, true, true, false>), grid, block, 0, stream, dv, num_steps,
                                         actions, act_stride, obs, rew, done);
            else if (loop_delta4) FS_LOOP(2, true);
            else FS_LOOP(2, false);
          }
        } else if (full) {
          last_kernel = "k_rollout_loop<FULL>";
          if (dv.env == FS_ENV_ACCEL)
            hipLaunchKernelGGL((fs::k_rollout_loop<0, true, true, MX>), grid, block, 0, stream, dv, num_steps, actions,
                               act_stride, obs, rew, done);
          else
            hipLaunchKernelGGL((fs::k_rollout_loop<1, true, true, MX>), grid, block, 0, stream, dv, num_steps, actions,
                               act_stride, obs, rew, done);
        }
        else if (dv.env == FS_ENV_ACCEL) { if (loop_delta4) FS_LOOP(0, true); else FS_LOOP(0, false); }
        else { if (loop_delta4) FS_LOOP(1, true); else FS_LOOP(1, false); }
#undef FS_LOOP
        HIP_TRY(hipGetLastError());
        return FS_OK;
      }
      }
    }
    // single-lane rings of IDM and RL vehicles (flowsim_ringrl.h): the RL experiments' populations and heads, masked
    // and zero-step launches included; the all-IDM AccelEnv rollout keeps its hand-written kernel below
    {
      const int f = dv.flags;
      const bool ma_head = dv.env == FS_ENV_WAVE_ATTENUATION_PO_MA || dv.env == FS_ENV_ACCEL_PO_MA;   // float32 only
      const bool ring_rl_ok = dv.nseg == 0 && !dv.junction_on && (f & fs::FLAG_IDM_SET) && !any_sim &&
                              !(f & fs::FLAG_HAS_FAILSAFE) && dv.sims_per_step == 1 && dv.integrator == FS_EULER &&
                              !dv.junction_mode && !dv.track_aux && !dv.sort_vehicles && dv.obs_perm == nullptr &&
                              !dv.evaluate &&
                              (dv.env == FS_ENV_ACCEL || dv.env == FS_ENV_WAVE_ATTENUATION_PO || (ma_head && !mixed)) &&
                              dv.N >= 2 && (dv.N % 2) == 0 && !force_generic && !no_ring_rl &&
                              (!(f & fs::FLAG_HAS_NOISE) || pair_noise || mixed);
      if (ring_rl_ok && !(pair_ok && (mixed || std::is_same<T, float>::value)) && (mixed || std::is_same<T, float>::value)) {
        const bool fast = ringrl_fast_ok();
        const int waves = (dv.R + (64 / ROW) - 1) / (64 / ROW);
        const dim3 grid((waves + 3) / 4), block(256);
        const bool po = dv.env == FS_ENV_WAVE_ATTENUATION_PO;
        // MC: the 16-step group form with several action columns (rows of 16 lanes only: 17..32 vehicles); the multi-agent
        // heads of that size always take it (it also steps without actions: the warm-up steps of a reset)
        const bool mc = ROW == 16 && std::is_same<T, float>::value && (ma_head || (actions != nullptr && dv.num_rl > 1));
        last_kernel = ma_head ? (dv.env == FS_ENV_ACCEL_PO_MA ? "k_ring_pair<AccelMA>" : "k_ring_pair<POMA>")
                              : (po ? "k_ring_pair<PO>" : "k_ring_pair<Accel>");
#define FS_RING(H_, NZ_, FA_, MC_)                                                                               \
  hipLaunchKernelGGL((fs::k_ring_pair<T, ROW, H_, NZ_, FA_, MC_>), grid, block, 0, stream, dv, num_steps, mask, actions, \
                     act_stride, obs, rew, done, obs_every_step)
#define FS_RING_F(H_, NZ_, MC_) do { if (fast) FS_RING(H_, NZ_, true, MC_); else FS_RING(H_, NZ_, false, MC_); } while (0)
#define FS_RING_N(H_, MC_) do { if (f & fs::FLAG_HAS_NOISE) FS_RING_F(H_, true, MC_); else FS_RING_F(H_, false, MC_); } while (0)
        if constexpr (std::is_same<T, float>::value) {
          if constexpr (ROW == 16) {
            if (ma_head) { if (dv.env == FS_ENV_ACCEL_PO_MA) FS_RING_N(3, true); else FS_RING_N(2, true); }
            else if (mc) { if (po) FS_RING_N(1, true); else FS_RING_N(0, true); }
            else { if (po) FS_RING_N(1, false); else FS_RING_N(0, false); }
          } else {
            if (ma_head) { if (dv.env == FS_ENV_ACCEL_PO_MA) FS_RING_N(3, false); else FS_RING_N(2, false); }
            else { if (po) FS_RING_N(1, false); else FS_RING_N(0, false); }
          }
        } else {
          if (po) FS_RING_N(1, false); else FS_RING_N(0, false);
        }
#undef FS_RING_N
#undef FS_RING_F
#undef FS_RING
        HIP_TRY(hipGetLastError());
        return FS_OK;
      }
    }
    if (mixed && num_steps == 0 && dv.nseg == 0) {       // observation of the current state (Env.reset)
      const int n = dv.R * dv.N;
      last_kernel = "k_obs_mixed";
      hipLaunchKernelGGL((fs::k_obs_mixed<T>), dim3((n + 255) / 256), dim3(256), 0, stream, dv, obs);
      HIP_TRY(hipGetLastError());
      return FS_OK;
    }
    if (mixed && !pair_ok && dv.nseg == 0)
      return fail(FS_ERR_UNSUPPORTED, "FS_MIXED: this launch fits neither mixed kernel (k_rollout_pair / k_ring_pair: "
                                      "single-lane ring, even number of IDM / RL vehicles, AccelEnv or "
                                      "WaveAttenuationPOEnv, track_aux = 0)");
    if (pair_ok && (mixed || std::is_same<T, float>::value)) {
      const bool fd = fastdiv_ok();
      const int waves = (dv.R + (64 / ROW) - 1) / (64 / ROW);
      const int wpb = pair_block / 64;
      const dim3 grid((waves + wpb - 1) / wpb), block(pair_block);
      const bool bc = neg_speed_possible;
#define FS_PAIR(D4, FD, BC)                                                                                   \
  hipLaunchKernelGGL((fs::k_rollout_pair<T, ROW, D4, FD, BC>), grid, block, 0, stream, dv, num_steps, obs, rew, done)
      last_kernel = speed_mode_any ? "k_rollout_pair+speed_mode" : "k_rollout_pair";
      if (dv.flags & fs::FLAG_HAS_NOISE) {
        if constexpr (std::is_same<T, float>::value) {
          last_kernel = speed_mode_any ? "k_rollout_pair+speed_mode+noise" : "k_rollout_pair+noise";
#define FS_PAIR_N(D4, FD, BC, SM_)                                                                            \
  hipLaunchKernelGGL((fs::k_rollout_pair<float, ROW, D4, FD, BC, SM_, true>), grid, block, 0, stream, dv, num_steps,  \
                     obs, rew, done)
          if (delta4 && fd && !bc) { if (speed_mode_any) FS_PAIR_N(true, true, false, true); else FS_PAIR_N(true, true, false, false); }
          else { if (speed_mode_any) FS_PAIR_N(false, false, true, true); else FS_PAIR_N(false, false, true, false); }
#undef FS_PAIR_N
        }
      } else if (speed_mode_any) {                   // the reference's default speed mode "right_of_way" lands here
        if (delta4 && fd && !bc)
          hipLaunchKernelGGL((fs::k_rollout_pair<T, ROW, true, true, false, true>), grid, block, 0, stream, dv,
                             num_steps, obs, rew, done);
        else                                  // any exponent, IEEE divisions, v < -100 check: always valid
          hipLaunchKernelGGL((fs::k_rollout_pair<T, ROW, false, false, true, true>), grid, block, 0, stream, dv,
                             num_steps, obs, rew, done);
      }
      else if (delta4 && fd) { if (bc) FS_PAIR(true, true, true); else FS_PAIR(true, true, false); }
      else if (delta4) { if (bc) FS_PAIR(true, false, true); else FS_PAIR(true, false, false); }
      else { if (bc) FS_PAIR(false, false, true); else FS_PAIR(false, false, false); }
#undef FS_PAIR
    } else if (fast_ok(mask, num_steps) && obs_every_step && dv.N > 1 && actions == nullptr &&
        size_t(dv.R) * 2 * dv.N * sizeof(float) < (size_t(1) << 32)) {   // 32-bit byte offsets inside one step's block
      const bool fd = fastdiv_ok();
      const int waves = blocks;                                   // one wave per 64/SEG replicas
      const int wpb = rollout_block / 64;                         // waves per block
      const dim3 grid((waves + wpb - 1) / wpb), block(rollout_block);
      const bool bc = neg_speed_possible;
      last_kernel = "k_rollout_idm";
#define FS_ROLLOUT(D4, FD, BC)                                                                               \
  hipLaunchKernelGGL((fs::k_rollout_idm<T, SEG, D4, FD, BC>), grid, block, 0, stream, dv, num_steps, obs, rew, \
                     done, d_dump)
      if (delta4 && fd) { if (bc) FS_ROLLOUT(true, true, true); else FS_ROLLOUT(true, true, false); }
      else if (delta4) { if (bc) FS_ROLLOUT(true, false, true); else FS_ROLLOUT(true, false, false); }
      else { if (bc) FS_ROLLOUT(false, false, true); else FS_ROLLOUT(false, false, false); }
#undef FS_ROLLOUT
    } else if (fast_ok(mask, num_steps)) {
      last_kernel = "k_steps<FAST>";
      hipLaunchKernelGGL((fs::k_steps<T, SEG, 1>), dim3(blocks), dim3(64), 0, stream, dv, num_steps, mask, actions,
                         act_stride, obs, rew, done, obs_every_step);
    } else if ((dv.flags & fs::FLAG_IDM_SET) && !force_generic) {
      last_kernel = "k_steps<CSET>";
      hipLaunchKernelGGL((fs::k_steps<T, SEG, 0, 1>), dim3(blocks), dim3(64), 0, stream, dv, num_steps, mask, actions,
                         act_stride, obs, rew, done, obs_every_step);
    } else {
      last_kernel = "k_steps";
      hipLaunchKernelGGL((fs::k_steps<T, SEG, 0, 0>), dim3(blocks), dim3(64), 0, stream, dv, num_steps, mask, actions,
                         act_stride, obs, rew, done, obs_every_step);
    }
    HIP_TRY(hipGetLastError());
    return FS_OK;
  }

  template <typename T>
  int Sim<T>::launch_policy_row16(const fs_policy* pol, int num_steps, int reset_done, const float* obs_in, float* obs,
                                  float* act, float* logp, float* rew, uint8_t* done) {
    fs::PolicyView pv;
    pv.w = pol->weights_dev;
    pv.log_std = pol->log_std_dev;
    pv.ctr = d_pol_ctr;
    pv.in_dim = pol->obs_dim;
    pv.num_hidden = pol->num_hidden;
    pv.n_out = pol->log_std_dev ? 1 : 2;
    pv.seed_lo = uint32_t(pol->seed & 0xFFFFFFFFull);
    pv.seed_hi = uint32_t(pol->seed >> 32);
    if (obs == nullptr) {                                  // eager: the policy alone
      const int rows = dv.R, blocks = (rows * 16 + 255) / 256;
      last_kernel = "k_policy_act";
      hipLaunchKernelGGL(fs::k_policy_act<16>, dim3(blocks), dim3(256), 0, stream, pv, dv.R, dv.rep0, obs_in, act, logp);
      HIP_TRY(hipGetLastError());
      return FS_OK;
    }
    if constexpr (std::is_same<T, float>::value || std::is_same<T, double>::value) {
      const bool fast = ringrl_fast_ok();
      const int waves = (dv.R + 3) / 4;
      const dim3 grid((waves + 3) / 4), block(256);
      const int wu = cfg.warmup_steps;
      last_kernel = "k_ring_policy";
#define FS_POL(NZ_, FA_)                                                                                          \
  hipLaunchKernelGGL((fs::k_ring_policy<T, NZ_, FA_>), grid, block, 0, stream, dv, pv, num_steps, reset_done, wu, obs, \
                     act, logp, rew, done)
      if constexpr (std::is_same<T, float>::value) {
        if (dv.flags & fs::FLAG_HAS_NOISE) { if (fast) FS_POL(true, true); else FS_POL(true, false); }
        else { if (fast) FS_POL(false, true); else FS_POL(false, false); }
      } else {
        if (dv.flags & fs::FLAG_HAS_NOISE) { if (fast) FS_POL(true, true); else FS_POL(true, false); }
        else { if (fast) FS_POL(false, true); else FS_POL(false, false); }
      }
#undef FS_POL
      HIP_TRY(hipGetLastError());
    }
    return FS_OK;
  }

  template <typename T>
  int Sim<T>::launch_policy_loop16(const fs_policy* pol, int num_steps, int reset_done, const float* obs_in, float* obs,
                                   float* act, float* logp, float* rew, uint8_t* done) {
    fs::PolicyView pv;
    pv.w = pol->weights_dev;
    pv.log_std = pol->log_std_dev;
    pv.ctr = d_pol_ctr;
    pv.in_dim = pol->obs_dim;
    pv.num_hidden = pol->num_hidden;
    pv.n_out = pol->log_std_dev ? 1 : 2;
    pv.seed_lo = uint32_t(pol->seed & 0xFFFFFFFFull);
    pv.seed_hi = uint32_t(pol->seed >> 32);
    if (obs == nullptr) {                                  // eager: the policy alone
      const int blocks = (dv.R * 16 + 255) / 256;
      last_kernel = "k_policy_act";
      hipLaunchKernelGGL(fs::k_policy_act<16>, dim3(blocks), dim3(256), 0, stream, pv, dv.R, dv.rep0, obs_in, act, logp);
      HIP_TRY(hipGetLastError());
      return FS_OK;
    }
    if constexpr (std::is_same<T, float>::value) {
      const int waves = (dv.R + 3) / 4;
      const dim3 grid((waves + 3) / 4), block(256);
      const bool fastc = loop_delta4 && loop_fastc_ok();
      last_kernel = "k_loop_policy";
#define FS_LPOL(H_, D4_, FC_)                                                                                     \
  hipLaunchKernelGGL((fs::k_loop_policy<H_, D4_, FC_>), grid, block, 0, stream, dv, pv, num_steps, reset_done, obs, act, \
                     logp, rew, done)
      const bool po = dv.env == FS_ENV_WAVE_ATTENUATION_PO;
      if (po) { if (fastc) FS_LPOL(1, true, true); else if (loop_delta4) FS_LPOL(1, true, false); else FS_LPOL(1, false, false); }
      else { if (fastc) FS_LPOL(0, true, true); else if (loop_delta4) FS_LPOL(0, true, false); else FS_LPOL(0, false, false); }
#undef FS_LPOL
      HIP_TRY(hipGetLastError());
      return FS_OK;
    } else {
      return fail(FS_ERR_UNSUPPORTED, "k_loop_policy is a float32 kernel");
    }
  }

}  // namespace fsim
