#!/usr/bin/env python3
"""Generates valu_issue.hip: issue cost of the VALU instructions the rollout kernels are made of (gfx950).

Per instruction: a loop body of 64 copies over 8 destination registers (round robin, so copies 8 apart are
dependent: latency of 8 issue slots is hidden), ITER iterations, WPS waves per SIMD over the whole chip.
Output: cycles per wave-instruction as one wave sees them (s_memtime) and per SIMD (= that / WPS).

    python gen_valu_issue.py > valu_issue.hip && hipcc --offload-arch=gfx950 -O3 -o valu_issue valu_issue.hip
"""
D32 = lambda i: "v%d" % (10 + i)
S32 = lambda i: "v%d" % (20 + i)
D64 = lambda i: "v[%d:%d]" % (30 + 2 * i, 31 + 2 * i)
S64 = lambda i: "v[%d:%d]" % (50 + 2 * i, 51 + 2 * i)

OPS = [
    ("v_fma_f32", lambda i: "v_fma_f32 %s, %s, %s, %s" % (D32(i), S32(i), S32(i), D32(i))),
    ("v_add_f32", lambda i: "v_add_f32 %s, %s, %s" % (D32(i), S32(i), D32(i))),
    ("v_mul_f32", lambda i: "v_mul_f32 %s, %s, %s" % (D32(i), S32(i), D32(i))),
    ("v_max_f32", lambda i: "v_max_f32 %s, %s, %s" % (D32(i), S32(i), D32(i))),
    ("v_pk_fma_f32", lambda i: "v_pk_fma_f32 %s, %s, %s, %s" % (D64(i), S64(i), S64(i), D64(i))),
    ("v_pk_mul_f32", lambda i: "v_pk_mul_f32 %s, %s, %s" % (D64(i), S64(i), D64(i))),
    ("v_pk_add_f32", lambda i: "v_pk_add_f32 %s, %s, %s" % (D64(i), S64(i), D64(i))),
    ("v_pk_add_f32 op_sel", lambda i: "v_pk_add_f32 %s, %s, %s op_sel:[1,0] op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1]"
     % (D64(i), S64(i), D64(i))),
    ("v_pk_mov_b32", lambda i: "v_pk_mov_b32 %s, %s, %s" % (D64(i), S64(i), D64(i))),
    ("v_fma_f64", lambda i: "v_fma_f64 %s, %s, %s, %s" % (D64(i), S64(i), S64(i), D64(i))),
    ("v_add_f64", lambda i: "v_add_f64 %s, %s, %s" % (D64(i), S64(i), D64(i))),
    ("v_mul_f64", lambda i: "v_mul_f64 %s, %s, %s" % (D64(i), S64(i), D64(i))),
    ("v_max_f64", lambda i: "v_max_f64 %s, %s, %s" % (D64(i), S64(i), D64(i))),
    ("v_cvt_f32_f64", lambda i: "v_cvt_f32_f64 %s, %s" % (D32(i), S64(i))),
    ("v_cvt_f64_f32", lambda i: "v_cvt_f64_f32 %s, %s" % (D64(i), S32(i))),
    ("v_rcp_f32", lambda i: "v_rcp_f32 %s, %s" % (D32(i), S32(i))),
    ("v_sqrt_f32", lambda i: "v_sqrt_f32 %s, %s" % (D32(i), S32(i))),
    ("v_rcp_f64", lambda i: "v_rcp_f64 %s, %s" % (D64(i), S64(i))),
    ("v_mov_b32", lambda i: "v_mov_b32 %s, %s" % (D32(i), S32(i))),
    ("v_mov_b32 dpp row_shl:1", lambda i: "v_mov_b32_dpp %s, %s row_shl:1 row_mask:0xf bank_mask:0xf" % (D32(i), S32(i))),
    ("v_mov_b32 dpp row_newbcast:0", lambda i: "v_mov_b32_dpp %s, %s row_newbcast:0 row_mask:0xf bank_mask:0xf" % (D32(i), S32(i))),
    ("v_mov_b32 dpp wave_shl:1", lambda i: "v_mov_b32_dpp %s, %s wave_shl:1 row_mask:0xf bank_mask:0xf" % (D32(i), S32(i))),
    ("v_add_f32 dpp row_shl:1", lambda i: "v_add_f32_dpp %s, %s, %s row_shl:1 row_mask:0xf bank_mask:0xf" % (D32(i), S32(i), D32(i))),
    ("v_cndmask_b32", lambda i: "v_cndmask_b32 %s, %s, %s, vcc" % (D32(i), S32(i), D32(i))),
    ("v_cndmask_b32 dpp", lambda i: "v_cndmask_b32_dpp %s, %s, %s, vcc row_newbcast:0 row_mask:0xf bank_mask:0xf" % (D32(i), S32(i), D32(i))),
    ("v_cmp_lt_f32", lambda i: "v_cmp_lt_f32 vcc, %s, %s" % (S32(i), D32(i))),
    ("v_cmp_lt_f64", lambda i: "v_cmp_lt_f64 vcc, %s, %s" % (S64(i), D64(i))),
    ("v_cmp_lt_f32 sgpr", lambda i: "v_cmp_lt_f32 s[%d:%d], %s, %s" % (20 + 2 * (i % 4), 21 + 2 * (i % 4), S32(i), D32(i))),
    ("v_readlane_b32", lambda i: "v_readlane_b32 s%d, %s, 5" % (20 + i, S32(i))),
    ("v_readfirstlane_b32", lambda i: "v_readfirstlane_b32 s%d, %s" % (20 + i, S32(i))),
    ("v_permlane16_swap", lambda i: "v_permlane16_swap_b32 %s, %s" % (D32(i), D32((i + 4) % 8))),
    ("v_and_b32", lambda i: "v_and_b32 %s, %s, %s" % (D32(i), S32(i), D32(i))),
    ("v_or_b32", lambda i: "v_or_b32 %s, %s, %s" % (D32(i), S32(i), D32(i))),
    ("v_lshl_add_u64", lambda i: "v_lshl_add_u64 %s, %s, 2, %s" % (D64(i), S64(i), D64(i))),
    ("v_add_u32", lambda i: "v_add_u32 %s, %s, %s" % (D32(i), S32(i), D32(i))),
    ("v_mul_lo_u32", lambda i: "v_mul_lo_u32 %s, %s, %s" % (D32(i), S32(i), D32(i))),
    ("v_mul_hi_u32", lambda i: "v_mul_hi_u32 %s, %s, %s" % (D32(i), S32(i), D32(i))),
    ("v_div_scale_f32", lambda i: "v_div_scale_f32 %s, vcc, %s, %s, %s" % (D32(i), S32(i), S32(i), D32(i))),
    ("v_div_fmas_f32", lambda i: "v_div_fmas_f32 %s, %s, %s, %s" % (D32(i), S32(i), S32(i), D32(i))),
    ("v_div_fixup_f32", lambda i: "v_div_fixup_f32 %s, %s, %s, %s" % (D32(i), S32(i), S32(i), D32(i))),
    ("v_floor_f32", lambda i: "v_floor_f32 %s, %s" % (D32(i), S32(i))),
    ("v_exp_f32", lambda i: "v_exp_f32 %s, %s" % (D32(i), S32(i))),
    ("v_log_f32", lambda i: "v_log_f32 %s, %s" % (D32(i), S32(i))),
    ("v_sin_f32", lambda i: "v_sin_f32 %s, %s" % (D32(i), S32(i))),
    ("ds_bpermute_b32", lambda i: "ds_bpermute_b32 %s, %s, %s" % (D32(i), S32(i), S32((i + 1) % 8))),
    ("s_nop 0", lambda i: "s_nop 0"),
    ("mix fma+pk_fma", lambda i: ("v_fma_f32 %s, %s, %s, %s" % (D32(i), S32(i), S32(i), D32(i))) if i % 2 == 0
     else ("v_pk_fma_f32 %s, %s, %s, %s" % (D64(i), S64(i), S64(i), D64(i)))),
    ("mix fma_f32+fma_f64", lambda i: ("v_fma_f32 %s, %s, %s, %s" % (D32(i), S32(i), S32(i), D32(i))) if i % 2 == 0
     else ("v_fma_f64 %s, %s, %s, %s" % (D64(i), S64(i), S64(i), D64(i)))),

    ("CHAIN v_fma_f32", lambda i: "v_fma_f32 v10, v20, v20, v10"),
    ("CHAIN v_pk_fma_f32", lambda i: "v_pk_fma_f32 v[30:31], v[50:51], v[50:51], v[30:31]"),
    ("CHAIN v_pk_mul_f32", lambda i: "v_pk_mul_f32 v[30:31], v[50:51], v[30:31]"),
    ("CHAIN v_add_f64", lambda i: "v_add_f64 v[30:31], v[50:51], v[30:31]"),
    ("CHAIN v_max_f32", lambda i: "v_max_f32 v10, v20, v10"),
    ("CHAIN v_rcp_f32", lambda i: "v_rcp_f32 v10, v10"),
    ("CHAIN v_mov_dpp row_shl:1", lambda i: "v_mov_b32_dpp v10, v10 row_shl:1 row_mask:0xf bank_mask:0xf"),
    ("CHAIN v_add_f32_dpp", lambda i: "v_add_f32_dpp v10, v10, v10 row_shl:1 row_mask:0xf bank_mask:0xf"),
    ("CHAIN fma -> dpp mov alternating", lambda i: "v_fma_f32 v10, v20, v20, v11" if i % 2 == 0 else "v_mov_b32_dpp v11, v10 row_shl:1 row_mask:0xf bank_mask:0xf"),
    ("CHAIN2 v_pk_fma_f32 (2 chains)", lambda i: "v_pk_fma_f32 v[%d:%d], v[50:51], v[50:51], v[%d:%d]" % (30 + 2 * (i % 2), 31 + 2 * (i % 2), 30 + 2 * (i % 2), 31 + 2 * (i % 2))),
    ("CHAIN4 v_pk_fma_f32 (4 chains)", lambda i: "v_pk_fma_f32 v[%d:%d], v[50:51], v[50:51], v[%d:%d]" % (30 + 2 * (i % 4), 31 + 2 * (i % 4), 30 + 2 * (i % 4), 31 + 2 * (i % 4))),
    ("CHAIN2 v_fma_f32 (2 chains)", lambda i: "v_fma_f32 v%d, v20, v20, v%d" % (10 + i % 2, 10 + i % 2)),
    ("v_cndmask_b32 e64 sgpr", lambda i: "v_cndmask_b32_e64 %s, %s, %s, s[28:29]" % (D32(i), S32(i), D32(i))),
    ("cmp+cndmask pairs", lambda i: ("v_cmp_lt_f32 vcc, %s, %s" % (S32(i), S32((i + 1) % 8))) if i % 2 == 0 else ("v_cndmask_b32 %s, %s, %s, vcc" % (D32(i), S32(i), D32(i)))),
    ("cmp_sgpr+cndmask_e64 pairs", lambda i: ("v_cmp_lt_f32 s[28:29], %s, %s" % (S32(i), S32((i + 1) % 8))) if i % 2 == 0 else ("v_cndmask_b32_e64 %s, %s, %s, s[28:29]" % (D32(i), S32(i), D32(i)))),
    ("v_max_f32 e64 abs", lambda i: "v_max_f32_e64 %s, |%s|, %s" % (D32(i), S32(i), D32(i))),
    ("v_min_f32", lambda i: "v_min_f32 %s, %s, %s" % (D32(i), S32(i), D32(i))),
    ("v_med3_f32", lambda i: "v_med3_f32 %s, %s, %s, %s" % (D32(i), S32(i), S32(i), D32(i))),
    ("v_sub_f32", lambda i: "v_sub_f32 %s, %s, %s" % (D32(i), S32(i), D32(i))),
    ("v_fmac_f32", lambda i: "v_fmac_f32 %s, %s, %s" % (D32(i), S32(i), S32(i))),
    ("v_mul_f32 e64 neg", lambda i: "v_mul_f32_e64 %s, -%s, %s" % (D32(i), S32(i), D32(i))),
    ("v_fma_f32 sgpr operand", lambda i: "v_fma_f32 %s, %s, s30, %s" % (D32(i), S32(i), D32(i))),
    ("v_ashrrev_i32", lambda i: "v_ashrrev_i32 %s, 31, %s" % (D32(i), D32(i))),
    ("v_bfi_b32", lambda i: "v_bfi_b32 %s, %s, %s, %s" % (D32(i), S32(i), S32(i), D32(i))),
    ("v_cvt_f32_u32", lambda i: "v_cvt_f32_u32 %s, %s" % (D32(i), S32(i))),
    ("v_cvt_u32_f32", lambda i: "v_cvt_u32_f32 %s, %s" % (D32(i), S32(i))),
    ("v_sub_u32", lambda i: "v_sub_u32 %s, %s, %s" % (D32(i), S32(i), D32(i))),
    ("v_lshlrev_b32", lambda i: "v_lshlrev_b32 %s, 3, %s" % (D32(i), D32(i))),
    ("v_add3_u32", lambda i: "v_add3_u32 %s, %s, %s, %s" % (D32(i), S32(i), S32(i), D32(i))),
    ("v_lshl_add_u32", lambda i: "v_lshl_add_u32 %s, %s, 2, %s" % (D32(i), S32(i), D32(i))),
    ("v_min_u32", lambda i: "v_min_u32 %s, %s, %s" % (D32(i), S32(i), D32(i))),
    ("v_cmp_lt_u32", lambda i: "v_cmp_lt_u32 vcc, %s, %s" % (S32(i), D32(i))),
    ("v_permlane32_swap", lambda i: "v_permlane32_swap_b32 %s, %s" % (D32(i), D32((i + 4) % 8))),
    ("v_mad_u64_u32", lambda i: "v_mad_u64_u32 %s, s[20:21], %s, %s, %s" % (D64(i), S32(i), S32((i + 1) % 8), S64(i))),
    ("v_mul_hi_u32 + v_mul_lo_u32 pair", lambda i: ("v_mul_hi_u32 %s, %s, %s" % (D32(i), S32(i), S32((i + 1) % 8))) if i % 2 == 0 else ("v_mul_lo_u32 %s, %s, %s" % (D32(i), S32(i - 1), S32(i)))),
    # control flow as a wave alone on its SIMD sees it (the rollout kernels' steps are branchy straight-line code)
    ("s_branch taken (over 1 instr)", lambda i: "s_branch 1f\\ns_nop 0\\n1:"),
    ("s_cbranch_vccnz taken (over 1)", lambda i: "s_cbranch_vccnz 1f\\ns_nop 0\\n1:"),
    ("s_cbranch_vccz not taken", lambda i: "s_cbranch_vccz 1f\\n1:"),
    ("v_cmp vcc + s_cbranch_vccz n.t.", lambda i: "v_cmp_lt_f32 vcc, v20, v21\\ns_cbranch_vccz 1f\\n1:"),
    ("v_cmp vcc + s_cbranch_vccnz taken", lambda i: "v_cmp_lt_f32 vcc, v21, v20\\ns_cbranch_vccnz 1f\\ns_nop 0\\n1:"),
    ("v_cmp sgpr + s_and_b64 on it", lambda i: "v_cmp_lt_f32 s[20:21], v20, v21\\ns_and_b64 s[22:23], s[20:21], s[28:29]"),
    ("v_cmp sgpr + s_and_b64 + v_cndmask on that", lambda i: "v_cmp_lt_f32 s[20:21], v20, v21\\ns_and_b64 s[22:23], s[20:21], s[28:29]\\nv_cndmask_b32_e64 v10, v20, v10, s[22:23]"),
    ("v_cmp x2 + s_and + cndmask (A & B ? .. : ..)", lambda i: "v_cmp_lt_f32 s[20:21], v20, v21\\nv_cmp_lt_f32 s[24:25], v21, v22\\ns_and_b64 s[22:23], s[20:21], s[24:25]\\nv_cndmask_b32_e64 v10, v20, v10, s[22:23]"),
    ("v_cmp, cndmask, v_cmp, cndmask (no SALU)", lambda i: "v_cmp_lt_f32 s[20:21], v20, v21\\nv_cndmask_b32_e64 v11, 0, 1, s[20:21]\\nv_cmp_lt_f32 s[24:25], v21, v22\\nv_cndmask_b32_e64 v10, 0, v11, s[24:25]"),
    ("v_readfirstlane + s_add on it", lambda i: "v_readfirstlane_b32 s20, v20\\ns_add_u32 s22, s20, s30"),
    ("s_branch taken over 16 instrs", lambda i: "s_branch 1f\\n" + "s_nop 0\\n" * 16 + "1:"),
]

CLOB = ", ".join('"v%d"' % r for r in list(range(10, 18)) + list(range(30, 46))) + \
    ', "vcc", "scc", ' + ", ".join('"s%d"' % r for r in range(20, 31))
ALLREG = ", ".join('"v%d"' % r for r in list(range(10, 18)) + list(range(20, 28)) + list(range(30, 46)) + list(range(50, 66)))

print("""// GENERATED by gen_valu_issue.py -- do not edit
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at line %d\\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef void (*kern_t)(int, unsigned long long*, float);
""")
init = []
for i in range(8):
    init.append("v_mov_b32 %s, %%0" % D32(i))
    init.append("v_mov_b32 %s, %%0" % S32(i))
    init.append("v_cvt_f64_f32 %s, %%0" % D64(i))
    init.append("v_cvt_f64_f32 %s, %%0" % S64(i))
init += ["s_mov_b64 s[28:29], 0x5555", "s_mov_b64 vcc, 0x3333", "s_mov_b32 s30, 0x3f800000"]
INIT = "\\n".join(init)
for n, (name, fn) in enumerate(OPS):
    body = "\\n".join(fn(i % 8) for i in range(64))
    ds = name.startswith("ds_")
    print("""__global__ __launch_bounds__(1024) void k%d(int iters, unsigned long long* out, float seed) {
  asm volatile("%s" :: "v"(seed) : %s, "vcc", "s28", "s29", "s30");
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    asm volatile("%s%s" ::: %s);
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x %% 64 == 0) out[(blockIdx.x * blockDim.x + threadIdx.x) / 64] = t1 - t0;
}""" % (n, INIT, ALLREG, body, "\\ns_waitcnt lgkmcnt(0)" if ds else "", CLOB))
print("struct Op { const char* name; kern_t fn; };")
print("static Op ops[] = {")
for n, (name, fn) in enumerate(OPS):
    print('  {"%s", k%d},' % (name, n))
print("};")
print("""
int main(int argc, char** argv) {
  int iters = 2000;
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  printf("device %s, %d CUs, clock %d kHz\\n", prop.name, cus, prop.clockRate);
  unsigned long long* d;
  CK(hipMalloc(&d, sizeof(unsigned long long) * cus * 32));
  std::vector<unsigned long long> h(cus * 32);
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  printf("%-32s %s\\n", "instruction", "waves/SIMD: cycles per wave-instr seen by a wave (s_memtime) | per SIMD from wall time @2.4GHz");
  for (auto& op : ops) {
    printf("%-32s", op.name);
    for (int wps : {1, 2, 4}) {
      const int block = 256 * wps;          // 4 SIMDs x wps waves
      op.fn<<<cus, block, 0, 0>>>(10, d, 1.0f);
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0, 0));
      op.fn<<<cus, block, 0, 0>>>(iters, d, 1.0f);
      CK(hipEventRecord(e1, 0));
      CK(hipDeviceSynchronize());
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      const int waves = cus * block / 64;
      CK(hipMemcpy(h.data(), d, sizeof(unsigned long long) * waves, hipMemcpyDeviceToHost));
      std::sort(h.begin(), h.begin() + waves);
      const double med = double(h[waves / 2]);
      const double per_wave = med / (double(iters) * 64);       // s_memtime ticks (100 MHz constant clock on gfx9? reported raw)
      const double per_simd = ms * 1e-3 * 2.4e9 / (double(iters) * 64 * wps);
      printf("  w%d: %6.2f | %5.2f", wps, per_wave, per_simd);
    }
    printf("\\n");
  }
  return 0;
}""")
