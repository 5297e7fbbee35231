// valu_calib.hip — calibrates the VALU issue ceiling the VALU-bound kernels are priced against.
//
// For each instruction kind and each requested occupancy it measures INSIDE the kernel, per wave: shader cycles
// (s_memtime), wall time (s_memrealtime, 100 MHz), and where the wave ran (HW_ID: XCC / SE / CU / SIMD).  From those:
//   * placement: how many CUs the launch really used and how many waves were co-resident per SIMD (time-weighted) —
//     a launch of CUs x k workgroups does NOT always land k waves on every SIMD;
//   * cycles per wave-instruction per SIMD = (span of the SIMD in shader cycles) / (instructions issued on that SIMD),
//     averaged over SIMDs — independent of the clock;
//   * the shader clock while the kernel ran (s_memtime delta / s_memrealtime delta x 100 MHz);
//   * from HIP events around the launch, wave-instructions per second of the whole chip.
// Build + run: scripts/valu_calib.sh (hipcc --offload-arch=gfx950, no other dependency).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <vector>

constexpr int UNROLL = 16;

#define OP(NAME, BODY)                                                                                   \
  struct NAME {                                                                                          \
    static constexpr const char *name = #NAME;                                                           \
    static __device__ __forceinline__ void go(float &x, float &y, float a, float b) { BODY; }            \
  };

namespace op {
// three VGPR sources (what valu_peak.hip measured)
OP(fma_vvv, asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b)))
OP(fma_vsv, asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "s"(a), "v"(b)))      // one SGPR source
OP(fma_vcc, asm volatile("v_fma_f32 %0, %0, 0.5, 1.0" : "+v"(x)))                      // inline constants
OP(fma_mods, asm volatile("v_fma_f32 %0, -%0, |%1|, %2" : "+v"(x) : "v"(a), "v"(b)))  // source modifiers
// VOP2 forms
OP(fmac_vv, asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(x) : "v"(a), "v"(b)))
OP(mul_vv, asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x) : "v"(a)))
OP(mul_sv, asm volatile("v_mul_f32 %0, %1, %0" : "+v"(x) : "s"(a)))
OP(mul_lv, asm volatile("v_mul_f32 %0, 0x3f7fbe77, %0" : "+v"(x)))                    // 32-bit literal
OP(mul_cv, asm volatile("v_mul_f32 %0, 0.5, %0" : "+v"(x)))                           // inline constant
OP(add_vv, asm volatile("v_add_f32 %0, %0, %1" : "+v"(x) : "v"(b)))
OP(sub_vv, asm volatile("v_sub_f32 %0, %0, %1" : "+v"(x) : "v"(b)))
OP(max_vv, asm volatile("v_max_f32 %0, %0, %1" : "+v"(x) : "v"(b)))
OP(min_vv, asm volatile("v_min_f32 %0, %0, %1" : "+v"(x) : "v"(b)))
OP(max_cv, asm volatile("v_max_f32 %0, 1.0, %0" : "+v"(x)))
OP(med3, asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b)))
OP(and_vv, asm volatile("v_and_b32 %0, %0, %1" : "+v"(x) : "v"(b)))
OP(addu_vv, asm volatile("v_add_u32 %0, %0, %1" : "+v"(x) : "v"(b)))
OP(cnd_vcc, asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x) : "v"(b)))
OP(cnd_sgpr, asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(x) : "v"(b) : "s20", "s21"))
OP(cmp_vcc, asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(x), "v"(b) : "vcc"))
OP(cmp_sgpr, asm volatile("v_cmp_lt_f32_e64 s[20:21], %0, %1" : : "v"(x), "v"(b) : "s20", "s21"))
OP(mov, asm volatile("v_mov_b32 %0, %1" : "+v"(x) : "v"(b)))
OP(mov_dpp, asm volatile("v_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(x)))
OP(mov_dpp_row, asm volatile("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(x)))
OP(add_dpp, asm volatile("v_add_f32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(x) : "v"(b)))
// round 6: integer max / min (floors of non-negative floats), three-operand max, bit-field insert, class compare, SDWA-free abs
OP(max_i32, asm volatile("v_max_i32 %0, %0, %1" : "+v"(x) : "v"(b)))
OP(max_u32, asm volatile("v_max_u32 %0, %0, %1" : "+v"(x) : "v"(b)))
OP(min_u32, asm volatile("v_min_u32 %0, %0, %1" : "+v"(x) : "v"(b)))
OP(max_i32_lit, asm volatile("v_max_i32 %0, 0x0da24260, %0" : "+v"(x)))
OP(max3_f32, asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b)))
OP(max3_u32, asm volatile("v_max3_u32 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b)))
OP(bfi, asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(x) : "v"(a), "v"(b)))
OP(cmp_class, asm volatile("v_cmp_class_f32 vcc, %0, %1" : : "v"(x), "v"(b) : "vcc"))
OP(cmp_u32, asm volatile("v_cmp_lt_u32 vcc, %0, %1" : : "v"(x), "v"(b) : "vcc"))
OP(sub_u32, asm volatile("v_sub_u32 %0, %0, %1" : "+v"(x) : "v"(b)))
OP(ashr, asm volatile("v_ashrrev_i32 %0, 31, %0" : "+v"(x)))
OP(and_or, asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b)))
OP(fmamk, asm volatile("v_fmamk_f32 %0, %0, 0x3f7fbe77, %1" : "+v"(x) : "v"(b)))
OP(fmaak, asm volatile("v_fmaak_f32 %0, %0, %1, 0x3f7fbe77" : "+v"(x) : "v"(b)))
OP(ldexp_f, asm volatile("v_ldexp_f32 %0, %0, %1" : "+v"(x) : "v"(b)))
OP(frexp_mant, asm volatile("v_frexp_mant_f32 %0, %0" : "+v"(x)))
// transcendental unit
OP(rcp, asm volatile("v_rcp_f32 %0, %0" : "+v"(x)))
OP(rsq, asm volatile("v_rsq_f32 %0, %0" : "+v"(x)))
OP(sqrt, asm volatile("v_sqrt_f32 %0, %0" : "+v"(x)))
OP(exp, asm volatile("v_exp_f32 %0, %0" : "+v"(x)))
OP(log, asm volatile("v_log_f32 %0, %0" : "+v"(x)))
// mixes: does the transcendental unit overlap the main pipe?
OP(mix_fma3_rcp1, asm volatile("v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %0, %0, %2, %3\n v_rcp_f32 %1, %1"
                               : "+v"(x), "+v"(y) : "v"(a), "v"(b)))
OP(mix_fma7_rcp1, asm volatile("v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %0, %0, %2, %3\n"
                               "v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %0, %0, %2, %3\n v_rcp_f32 %1, %1"
                               : "+v"(x), "+v"(y) : "v"(a), "v"(b)))
// a k_step-like mix: mul/add/fma/max/sub with VGPR operands, no transcendental
OP(mix_alu5, asm volatile("v_mul_f32 %0, %0, %2\n v_add_f32 %1, %1, %3\n v_fma_f32 %0, %0, %2, %3\n v_max_f32 %1, %1, %3\n v_sub_f32 %0, %0, %1"
                          : "+v"(x), "+v"(y) : "v"(a), "v"(b)))
OP(mix_alu5_sgpr, asm volatile("v_mul_f32 %0, %2, %0\n v_add_f32 %1, %3, %1\n v_fma_f32 %0, %0, %2, %1\n v_max_f32 %1, %3, %1\n v_sub_f32 %0, %0, %1"
                               : "+v"(x), "+v"(y) : "s"(a), "s"(b)))
// LDS pipe beside the VALU
OP(mix_fma4_dsr1, asm volatile("v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %0, %0, %2, %3\n"
                               "ds_read_b32 %1, %4\n s_waitcnt lgkmcnt(8)"
                               : "+v"(x), "+v"(y) : "v"(a), "v"(b), "v"((int)(threadIdx.x * 4))))
// packed fp32 (two floats per lane and instruction, operands in aligned VGPR pairs): does one instruction cost one issue slot?
typedef float v2f __attribute__((ext_vector_type(2)));
#define PK(NAME, INSTR)                                                                                  \
  struct NAME {                                                                                          \
    static constexpr const char *name = #NAME;                                                           \
    static __device__ __forceinline__ void go(float &x, float &y, float a, float b) {                    \
      v2f X, A2, B2;                                                                                     \
      X.x = x; X.y = y; A2.x = a; A2.y = a; B2.x = b; B2.y = b;                                          \
      asm volatile(INSTR : "+v"(X) : "v"(A2), "v"(B2));                                                  \
      x = X.x; y = X.y;                                                                                  \
    }                                                                                                    \
  };
PK(pk_fma, "v_pk_fma_f32 %0, %0, %1, %2")
PK(pk_mul, "v_pk_mul_f32 %0, %0, %1")
PK(pk_add, "v_pk_add_f32 %0, %0, %2")
PK(pk_fma_neg, "v_pk_fma_f32 %0, %0, %1, %2 neg_lo:[0,1,0] neg_hi:[0,1,0]")
PK(pk_fma_sel, "v_pk_fma_f32 %0, %0, %1, %2 op_sel:[0,1,0] op_sel_hi:[1,0,1]")
// 16-bit packed / dot-product forms (SPH range test on staged int16 or fp16 coordinates), bit scans, add-with-carry
OP(pk_sub_i16, asm volatile("v_pk_sub_i16 %0, %0, %1" : "+v"(x) : "v"(b)))
OP(dot2_i32_i16, asm volatile("v_dot2_i32_i16 %0, %1, %1, %0" : "+v"(x) : "v"(b)))
OP(dot2_u32_u16, asm volatile("v_dot2_u32_u16 %0, %1, %1, %0" : "+v"(x) : "v"(b)))
OP(pk_add_f16, asm volatile("v_pk_add_f16 %0, %0, %1" : "+v"(x) : "v"(b)))
OP(pk_mul_f16, asm volatile("v_pk_mul_f16 %0, %0, %1" : "+v"(x) : "v"(b)))
OP(dot2_f32_f16, asm volatile("v_dot2_f32_f16 %0, %1, %1, %0" : "+v"(x) : "v"(b)))
OP(mad_i32_i16, asm volatile("v_mad_i32_i16 %0, %1, %1, %0" : "+v"(x) : "v"(b)))
OP(addc, asm volatile("v_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(x) : : "vcc"))
OP(ffbl, asm volatile("v_ffbl_b32 %0, %1" : "+v"(x) : "v"(b)))
OP(bfrev, asm volatile("v_bfrev_b32 %0, %1" : "+v"(x) : "v"(b)))
OP(cnd_vcc_e64, asm volatile("v_cndmask_b32_e64 %0, %0, %1, vcc" : "+v"(x) : "v"(b)))
OP(cnd_vcc_set, asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(x) : "v"(b) : ))   // (run after vcc was written once: see k)
OP(mix_cmp_cnd_vcc, asm volatile("v_cmp_lt_f32_e32 vcc, %0, %2\n v_cndmask_b32_e32 %1, %1, %2, vcc" : "+v"(x), "+v"(y) : "v"(b) : "vcc"))
OP(mix_cmp_cnd_sgpr, asm volatile("v_cmp_lt_f32_e64 s[20:21], %0, %2\n v_cndmask_b32_e64 %1, %1, %2, s[20:21]" : "+v"(x), "+v"(y) : "v"(b) : "s20", "s21"))
OP(mix_cmp_fma_cnd_vcc, asm volatile("v_cmp_lt_f32_e32 vcc, %0, %2\n v_fma_f32 %0, %0, %2, %2\n v_fma_f32 %0, %0, %2, %2\n v_cndmask_b32_e32 %1, %1, %2, vcc" : "+v"(x), "+v"(y) : "v"(b) : "vcc"))
OP(mix_cmp_fma_cnd_sgpr, asm volatile("v_cmp_lt_f32_e64 s[20:21], %0, %2\n v_fma_f32 %0, %0, %2, %2\n v_fma_f32 %0, %0, %2, %2\n v_cndmask_b32_e64 %1, %1, %2, s[20:21]" : "+v"(x), "+v"(y) : "v"(b) : "s20", "s21"))
OP(mix_scan4, asm volatile("v_pk_sub_i16 %1, %2, %3\n v_dot2_i32_i16 %1, %1, %1, 0\n v_cmp_gt_i32 vcc, %2, %1\n v_addc_co_u32 %0, vcc, %0, %0, vcc"
                           : "+v"(x), "+v"(y) : "v"(a), "v"(b) : "vcc"))
OP(mix_scan6, asm volatile("v_sub_f32 %1, %2, %3\n v_sub_f32 %4, %3, %2\n v_mul_f32 %1, %1, %1\n v_fma_f32 %1, %4, %4, %1\n v_cmp_gt_f32 vcc, %2, %1\n v_addc_co_u32 %0, vcc, %0, %0, vcc"
                           : "+v"(x), "+v"(y) : "v"(a), "v"(b), "v"(y) : "vcc"))
}  // namespace op
using namespace op;

template <class O> struct per_call { static constexpr int n = 1; };
template <> struct per_call<mix_fma3_rcp1> { static constexpr int n = 4; };
template <> struct per_call<mix_fma7_rcp1> { static constexpr int n = 8; };
template <> struct per_call<mix_alu5> { static constexpr int n = 5; };
template <> struct per_call<mix_alu5_sgpr> { static constexpr int n = 5; };
template <> struct per_call<mix_fma4_dsr1> { static constexpr int n = 4; };   // VALU instructions only
template <> struct per_call<mix_scan4> { static constexpr int n = 4; };
template <> struct per_call<mix_cmp_cnd_vcc> { static constexpr int n = 2; };
template <> struct per_call<mix_cmp_cnd_sgpr> { static constexpr int n = 2; };
template <> struct per_call<mix_cmp_fma_cnd_vcc> { static constexpr int n = 4; };
template <> struct per_call<mix_cmp_fma_cnd_sgpr> { static constexpr int n = 4; };
template <> struct per_call<mix_scan6> { static constexpr int n = 6; };

struct Stamp { uint64_t cyc, w0, w1; uint32_t hwid, xcc; };

// DEP = 1: every instruction depends on the previous one of the same wave (latency chain); else 16 independent chains.
template <class O, int DEP> __global__ __launch_bounds__(256) void k(Stamp *stamps, float *sink, float a, float b, int iters) {
  __shared__ float lds[256];
  lds[threadIdx.x] = 1.f;
  float x[UNROLL], y[UNROLL];
#pragma unroll
  for (int i = 0; i < UNROLL; i++) { x[i] = threadIdx.x * 1e-3f + i + 1.f; y[i] = 1.f + i; }
  __builtin_amdgcn_s_barrier();
  const uint64_t w0 = wall_clock64();
  const uint64_t c0 = clock64();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < UNROLL; i++) {
      if (DEP) O::go(x[0], y[0], a, b);
      else O::go(x[i], y[i], a, b);
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)");
  const uint64_t c1 = clock64();
  const uint64_t w1 = wall_clock64();
  float s = lds[(threadIdx.x + 1) & 255];
#pragma unroll
  for (int i = 0; i < UNROLL; i++) s += x[i] + y[i];
  sink[blockIdx.x * 256 + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) {
    Stamp st;
    st.cyc = c1 - c0; st.w0 = w0; st.w1 = w1;
    st.hwid = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);    // HW_REG_HW_ID
    st.xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);     // HW_REG_XCC_ID
    stamps[blockIdx.x * 4 + (threadIdx.x >> 6)] = st;
  }
}

static int g_cus;
static Stamp *g_stamps;
static float *g_sink;

template <class O, int DEP> static void run(int wps, int iters) {
  const int blocks = g_cus * wps;   // one 256-thread workgroup = one wave on each of a CU's four SIMDs
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<O, DEP>), dim3(blocks), dim3(256), 0, 0, g_stamps, g_sink, 0.999f, 1e-3f, iters / 8);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<O, DEP>), dim3(blocks), dim3(256), 0, 0, g_stamps, g_sink, 0.999f, 1e-3f, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  const int nw = blocks * 4;
  std::vector<Stamp> st(nw);
  hipMemcpy(st.data(), g_stamps, st.size() * sizeof(Stamp), hipMemcpyDeviceToHost);
  const double ninstr = (double)iters * UNROLL * per_call<O>::n;
  double cyc = 0, wall = 0;
  uint64_t t0 = ~0ull, t1 = 0;
  struct Simd { uint64_t lo = ~0ull, hi = 0; double busy = 0; int n = 0; };
  std::map<uint32_t, Simd> simds;
  std::map<uint32_t, int> cus;
  for (const Stamp &s : st) {
    cyc += (double)s.cyc; wall += (double)(s.w1 - s.w0);
    t0 = std::min(t0, s.w0); t1 = std::max(t1, s.w1);
    const uint32_t cu = (s.xcc << 8) | ((s.hwid >> 8) & 0xff);        // XCC | SE, SH, CU
    const uint32_t sd = (cu << 2) | ((s.hwid >> 4) & 3);
    cus[cu]++;
    Simd &d = simds[sd];
    d.lo = std::min(d.lo, s.w0); d.hi = std::max(d.hi, s.w1); d.busy += (double)(s.w1 - s.w0); d.n++;
  }
  const double mhz = cyc / wall * 100.0;   // s_memrealtime ticks at 100 MHz
  double conc = 0, cpi = 0;
  for (auto &kv : simds) {
    const Simd &d = kv.second;
    const double span = (double)(d.hi - d.lo);
    conc += d.busy / span;
    cpi += span * (mhz / 100.0) / (d.n * ninstr);
  }
  conc /= simds.size(); cpi /= simds.size();
  const double rate = (double)nw * ninstr / (ms * 1e-3);
  printf("%-14s %s ask %d w/SIMD: %3zu CUs %4zu SIMDs, resident %.2f w/SIMD | %6.3f cyc/instr/SIMD | per-wave issue every %6.2f cyc | "
         "clock %4.0f MHz | span %6.3f ms, events %6.3f ms, %.3e wave-instr/s\n",
         O::name, DEP ? "dep  " : "indep", wps, cus.size(), simds.size(), conc, cpi, cyc / nw / ninstr, mhz,
         (double)(t1 - t0) * 1e-5, ms, rate);
  fflush(stdout);
}

template <class O> static void sweep(int iters) {
  for (int wps : {1, 2, 3, 4, 6, 8}) run<O, 0>(wps, iters);
}

int main(int argc, char **argv) {
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  g_cus = p.multiProcessorCount;
  const int iters = argc > 1 ? atoi(argv[1]) : 16384;
  hipMalloc(&g_stamps, (size_t)g_cus * 8 * 4 * sizeof(Stamp));
  hipMalloc(&g_sink, (size_t)g_cus * 8 * 256 * 4);
  printf("# %s, %d CUs, nominal clock %d MHz; %d x %d instructions per wave; workgroup = 256 threads = 1 wave per SIMD\n",
         p.gcnArchName, g_cus, p.clockRate / 1000, iters, UNROLL);
  if (argc > 2 && !strcmp(argv[2], "int16")) {   // only the 16-bit / bit-scan additions (round 4)
    sweep<fma_vvv>(iters);
    sweep<pk_sub_i16>(iters); sweep<dot2_i32_i16>(iters); sweep<dot2_u32_u16>(iters); sweep<pk_add_f16>(iters); sweep<pk_mul_f16>(iters);
    sweep<dot2_f32_f16>(iters); sweep<mad_i32_i16>(iters); sweep<addc>(iters); sweep<ffbl>(iters); sweep<bfrev>(iters);
    sweep<mix_scan4>(iters / 4); sweep<mix_scan6>(iters / 4);
    return 0;
  }
  if (argc > 2 && !strcmp(argv[2], "r06")) {     // round 6: candidates for replacing the half-rate v_max_f32 / v_cndmask floors
    for (int wps : {4, 6}) {
      run<fma_vvv, 0>(wps, iters); run<max_vv, 0>(wps, iters); run<max_i32, 0>(wps, iters); run<max_u32, 0>(wps, iters); run<min_u32, 0>(wps, iters);
      run<max_i32_lit, 0>(wps, iters); run<max3_f32, 0>(wps, iters); run<max3_u32, 0>(wps, iters); run<bfi, 0>(wps, iters);
      run<cmp_class, 0>(wps, iters); run<cmp_u32, 0>(wps, iters); run<sub_u32, 0>(wps, iters); run<ashr, 0>(wps, iters); run<and_or, 0>(wps, iters);
      run<fmamk, 0>(wps, iters); run<fmaak, 0>(wps, iters); run<ldexp_f, 0>(wps, iters); run<frexp_mant, 0>(wps, iters);
    }
    return 0;
  }
  if (argc > 2 && !strcmp(argv[2], "cnd")) {     // v_cndmask reading vcc (VOP2) against an SGPR pair (VOP3), alone and behind its compare
    sweep<cnd_vcc>(iters / 4); sweep<cnd_vcc_e64>(iters / 4); sweep<cnd_sgpr>(iters); sweep<mix_cmp_cnd_vcc>(iters / 2); sweep<mix_cmp_cnd_sgpr>(iters / 2);
    sweep<mix_cmp_fma_cnd_vcc>(iters / 4); sweep<mix_cmp_fma_cnd_sgpr>(iters / 4);
    return 0;
  }
  sweep<fma_vvv>(iters);
  sweep<fma_vsv>(iters);
  sweep<fma_vcc>(iters);
  sweep<fma_mods>(iters);
  sweep<fmac_vv>(iters);
  sweep<pk_fma>(iters);
  sweep<pk_mul>(iters);
  sweep<pk_add>(iters);
  sweep<pk_fma_neg>(iters);
  sweep<pk_fma_sel>(iters);
  sweep<mul_vv>(iters);
  sweep<mul_sv>(iters);
  sweep<mul_lv>(iters);
  sweep<mul_cv>(iters);
  sweep<add_vv>(iters);
  sweep<sub_vv>(iters);
  sweep<max_vv>(iters);
  sweep<min_vv>(iters);
  sweep<max_cv>(iters);
  sweep<med3>(iters);
  sweep<and_vv>(iters);
  sweep<addu_vv>(iters);
  sweep<cnd_vcc>(iters);
  sweep<cnd_sgpr>(iters);
  sweep<cmp_vcc>(iters);
  sweep<cmp_sgpr>(iters);
  sweep<mov>(iters);
  sweep<mov_dpp>(iters);
  sweep<mov_dpp_row>(iters);
  sweep<add_dpp>(iters);
  sweep<op::rcp>(iters / 2);
  sweep<op::rsq>(iters / 2);
  sweep<op::sqrt>(iters / 2);
  sweep<op::exp>(iters / 2);
  sweep<op::log>(iters / 2);
  sweep<mix_fma3_rcp1>(iters / 4);
  sweep<mix_fma7_rcp1>(iters / 8);
  sweep<mix_alu5>(iters / 4);
  sweep<mix_alu5_sgpr>(iters / 4);
  sweep<mix_fma4_dsr1>(iters / 4);
  // latency chains: one dependent stream per wave
  for (int wps : {1, 2, 3, 4, 8}) run<fma_vvv, 1>(wps, iters);
  for (int wps : {1, 2, 4}) run<mul_vv, 1>(wps, iters);
  for (int wps : {1, 2, 4}) run<op::rcp, 1>(wps, iters / 2);
  for (int wps : {1, 2, 4}) run<mov_dpp, 1>(wps, iters);
  hipFree(g_stamps); hipFree(g_sink);
  return 0;
}
