// How many cycles does a SIMD of gfx950 take to issue one wave64 integer VALU instruction?  (bench.py priced the
// tile kernel against 16 lanes per SIMD and clock -- 4 cycles per wave64 instruction; MI355X_MICROARCH.md says SIMD-32, 2 cycles, and
// holds a v_fma_f32 measurement for it.  The tile kernel's step is v_max_i32 / v_max3_i32 / v_add_u32 / v_cndmask_b32 / v_alignbit_b32 /
// v_xor / v_ffbl / DPP moves: none of them measured anywhere.)
//
// One workgroup per CU, W waves per SIMD (W = 1, 2, 4, 8: 4 W waves in the workgroup, the hardware deals them round-robin over the four SIMDs).
// Every wave runs N x 8 instructions of one kind on 8 independent registers (no chain shorter than 8 issue slots), bracketed by s_memtime;
// the figure per kind and W is  cycles per wave64 instruction per SIMD = (t_end_max - t_begin_min) / (N x 8 x W), in shader clocks (s_memtime
// counts at the constant 100 MHz reference on gfx9: the shader clock is taken from a v_fma_f32 reference loop... no: both are printed -- the
// ratio to the v_fma_f32 row, which the guide pins at 2 cycles, is the number to read).  A `dep` row per kind: one register, every instruction
// dependent on the one before (latency, not issue).
//
//   hipcc --offload-arch=gfx950 -O3 -o valu_issue scripts/micro/valu_issue.hip && ./valu_issue > profiles/r6_valu_issue.md
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>

enum Op { FMA_F32 = 0, MAX_I32, MAX3_I32, ADD_U32, ADD3_U32, CNDMASK, ALIGNBIT, XOR_B32, FFBL, MOV_DPP_ROWSHR, MOV_DPP_WAVESHR, MAX_DPP_ROWSHR, LSHL_ADD, MIN_U32, CMP_GT_I32, PK_MAX_I16, PK_ADD_U16, PERM_B32, BFE_U32, MAD_U32_U24, CNDMASK_VCC_SET, CNDMASK_E64, CNDMASK_CONST, AND_B32, OR_B32, LSHLREV, LSHRREV, SUB_U32, MOV_B32, CMP_E64, READLANE, SUBREV_CO, MAX_U32, MED3_I32, ADD_I32_CLAMP, PAIR_VCC, PAIR_SGPR, PAIR_VCC_NONOP, NOPS };
static const char* op_name[] = {"v_fma_f32", "v_max_i32", "v_max3_i32", "v_add_u32", "v_add3_u32", "v_cndmask_b32", "v_alignbit_b32", "v_xor_b32", "v_ffbl_b32",
                                "v_mov_b32 dpp row_shr:1", "v_mov_b32 dpp wave_shr:1", "v_max_i32 dpp row_shr:1", "v_lshl_add_u32", "v_min_u32", "v_cmp_gt_i32 (vcc)",
                                "v_pk_max_i16", "v_pk_add_u16", "v_perm_b32", "v_bfe_u32", "v_mad_u32_u24",
                                "v_cndmask_b32 (vcc set to 0x5555.. before the loop)", "v_cndmask_b32_e64 (mask in s[20:21])", "v_cndmask_b32_e64 v, 0, 1, s[20:21]", "v_and_b32", "v_or_b32", "v_lshlrev_b32", "v_lshrrev_b32", "v_sub_u32", "v_mov_b32", "v_cmp_gt_i32_e64 s[20:21]", "v_readlane_b32 (to s20)", "v_subrev_co_u32 (vcc)", "v_max_u32", "v_med3_i32", "v_add_i32 clamp",
                                "PAIR v_cmp_gt_i32 vcc + s_nop 1 + v_cndmask_b32_e32 vcc (per pair)", "PAIR v_cmp_gt_i32_e64 s[20:21] + s_nop 1 + v_cndmask_b32_e64 s[20:21] (per pair)", "PAIR v_cmp_gt_i32 vcc + v_cndmask_b32_e32 vcc on ANOTHER register's compare (per pair)"};

#define ONE(OPSTR, R) asm volatile(OPSTR : "+v"(R) : "v"(x), "v"(y) : "vcc", "s20", "s21");
#define EIGHT(OPSTR) ONE(OPSTR, r0) ONE(OPSTR, r1) ONE(OPSTR, r2) ONE(OPSTR, r3) ONE(OPSTR, r4) ONE(OPSTR, r5) ONE(OPSTR, r6) ONE(OPSTR, r7)
#define DEP8(OPSTR) ONE(OPSTR, r0) ONE(OPSTR, r0) ONE(OPSTR, r0) ONE(OPSTR, r0) ONE(OPSTR, r0) ONE(OPSTR, r0) ONE(OPSTR, r0) ONE(OPSTR, r0)

template <int OP, bool DEP>
__global__ __launch_bounds__(1024) void k(uint64_t* t_out, int* sink, int n) {
  int r0 = threadIdx.x, r1 = r0 + 1, r2 = r0 + 2, r3 = r0 + 3, r4 = r0 + 4, r5 = r0 + 5, r6 = r0 + 6, r7 = r0 + 7;
  int x = threadIdx.x * 3 + 1, y = threadIdx.x ^ 5;
  if (OP == CNDMASK_VCC_SET) asm volatile("s_mov_b64 vcc, 0x55555555" ::: "vcc");
  if (OP == CNDMASK_E64 || OP == CNDMASK_CONST) asm volatile("s_mov_b64 s[20:21], 0x55555555" ::: "s20", "s21");
  __syncthreads();
  const uint64_t t0 = __builtin_readcyclecounter();  // s_memtime
  const uint64_t w0 = wall_clock64();
  for (int i = 0; i < n; ++i) {
#define BODY(S) if (DEP) { DEP8(S) DEP8(S) DEP8(S) DEP8(S) } else { EIGHT(S) EIGHT(S) EIGHT(S) EIGHT(S) }
    if (OP == FMA_F32) { BODY("v_fma_f32 %0, %1, %2, %0") }
    else if (OP == MAX_I32) { BODY("v_max_i32 %0, %0, %1") }
    else if (OP == MAX3_I32) { BODY("v_max3_i32 %0, %0, %1, %2") }
    else if (OP == ADD_U32) { BODY("v_add_u32 %0, %0, %1") }
    else if (OP == ADD3_U32) { BODY("v_add3_u32 %0, %0, %1, %2") }
    else if (OP == CNDMASK) { BODY("v_cndmask_b32 %0, %0, %1, vcc") }
    else if (OP == ALIGNBIT) { BODY("v_alignbit_b32 %0, %0, %1, %2") }
    else if (OP == XOR_B32) { BODY("v_xor_b32 %0, %0, %1") }
    else if (OP == FFBL) { BODY("v_ffbl_b32 %0, %0") }
    else if (OP == MOV_DPP_ROWSHR) { BODY("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf") }
    else if (OP == MOV_DPP_WAVESHR) { BODY("v_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf") }
    else if (OP == MAX_DPP_ROWSHR) { BODY("v_max_i32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf") }
    else if (OP == LSHL_ADD) { BODY("v_lshl_add_u32 %0, %0, 1, %1") }
    else if (OP == MIN_U32) { BODY("v_min_u32 %0, %0, %1") }
    else if (OP == CMP_GT_I32) { BODY("v_cmp_gt_i32 vcc, %0, %1") }
    else if (OP == PK_MAX_I16) { BODY("v_pk_max_i16 %0, %0, %1") }
    else if (OP == PK_ADD_U16) { BODY("v_pk_add_u16 %0, %0, %1") }
    else if (OP == PERM_B32) { BODY("v_perm_b32 %0, %0, %1, %2") }
    else if (OP == BFE_U32) { BODY("v_bfe_u32 %0, %0, 1, 31") }
    else if (OP == MAD_U32_U24) { BODY("v_mad_u32_u24 %0, %0, %1, %2") }
    else if (OP == CNDMASK_VCC_SET) { BODY("v_cndmask_b32 %0, %0, %1, vcc") }
    else if (OP == CNDMASK_E64) { BODY("v_cndmask_b32_e64 %0, %0, %1, s[20:21]") }
    else if (OP == CNDMASK_CONST) { BODY("v_cndmask_b32_e64 %0, 0, 1, s[20:21]") }
    else if (OP == AND_B32) { BODY("v_and_b32 %0, %0, %1") }
    else if (OP == OR_B32) { BODY("v_or_b32 %0, %0, %1") }
    else if (OP == LSHLREV) { BODY("v_lshlrev_b32 %0, 1, %0") }
    else if (OP == LSHRREV) { BODY("v_lshrrev_b32 %0, 1, %0") }
    else if (OP == SUB_U32) { BODY("v_sub_u32 %0, %0, %1") }
    else if (OP == MOV_B32) { BODY("v_mov_b32 %0, %1") }
    else if (OP == CMP_E64) { BODY("v_cmp_gt_i32_e64 s[20:21], %0, %1") }
    else if (OP == READLANE) { BODY("v_readlane_b32 s20, %0, 3") }
    else if (OP == SUBREV_CO) { BODY("v_subrev_co_u32 %0, vcc, %0, %1") }
    else if (OP == MAX_U32) { BODY("v_max_u32 %0, %0, %1") }
    else if (OP == MED3_I32) { BODY("v_med3_i32 %0, %0, %1, %2") }
    else if (OP == PAIR_VCC) { BODY("v_cmp_gt_i32 vcc, %0, %1\n\ts_nop 1\n\tv_cndmask_b32 %0, %0, %2, vcc") }
    else if (OP == PAIR_SGPR) { BODY("v_cmp_gt_i32_e64 s[20:21], %0, %1\n\ts_nop 1\n\tv_cndmask_b32_e64 %0, %0, %2, s[20:21]") }
    else if (OP == PAIR_VCC_NONOP) { BODY("v_cndmask_b32 %0, %0, %2, vcc\n\tv_cmp_gt_i32 vcc, %0, %1") }
    else if (OP == ADD_I32_CLAMP) { BODY("v_add_i32 %0, %0, %1 clamp") }
  }
  asm volatile("s_nop 0" ::: "memory");
  const uint64_t t1 = __builtin_readcyclecounter();
  const uint64_t w1 = wall_clock64();
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if ((threadIdx.x & 63) == 0) { t_out[wave * 4 + 0] = t0; t_out[wave * 4 + 1] = t1; t_out[wave * 4 + 2] = w0; t_out[wave * 4 + 3] = w1; }
  sink[blockIdx.x * blockDim.x + threadIdx.x] = r0 ^ r1 ^ r2 ^ r3 ^ r4 ^ r5 ^ r6 ^ r7;
}

struct Res { double cyc_memtime, ns_wall; };

template <int OP, bool DEP>
Res run(int W, int blocks, int n, uint64_t* d_t, int* d_sink) {
  // W = 8 is two workgroups of 1024 threads per CU (a workgroup holds at most 16 waves): twice the workgroups, each one's span counted
  // against all 8 waves per SIMD (the dispatcher places the two side by side; the kernel needs < 16 registers and no LDS)
  const int threads = W == 8 ? 1024 : 64 * 4 * W;
  if (W == 8) blocks *= 2;
  const int nw = blocks * threads / 64;
  std::vector<uint64_t> h((size_t)nw * 4);
  Res best{1e30, 1e30};
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL((k<OP, DEP>), dim3(blocks), dim3(threads), 0, 0, d_t, d_sink, n);
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(h.data(), d_t, h.size() * 8, hipMemcpyDeviceToHost);
    // per workgroup (= per CU): the span from the first wave's start to the last wave's end; the median over the workgroups
    std::vector<double> cy, ns;
    const int wpb = threads / 64;
    for (int b = 0; b < blocks; ++b) {
      uint64_t a0 = ~0ull, a1 = 0, b0 = ~0ull, b1 = 0;
      for (int w = 0; w < wpb; ++w) {
        const uint64_t* q = &h[((size_t)b * wpb + w) * 4];
        a0 = std::min(a0, q[0]); a1 = std::max(a1, q[1]); b0 = std::min(b0, q[2]); b1 = std::max(b1, q[3]);
      }
      cy.push_back((double)(a1 - a0)); ns.push_back((double)(b1 - b0) * 10.0);  // s_memrealtime: 100 MHz
    }
    std::sort(cy.begin(), cy.end()); std::sort(ns.begin(), ns.end());
    const double per = (double)n * 32.0 * (double)W;
    best.cyc_memtime = std::min(best.cyc_memtime, cy[cy.size() / 2] / per);
    best.ns_wall = std::min(best.ns_wall, ns[ns.size() / 2] / per);
  }
  return best;
}

template <int OP>
void row(uint64_t* d_t, int* d_sink, int blocks, int n) {
  printf("| %s |", op_name[OP]);
  double ns1 = 0;
  for (int W : {1, 2, 4, 8}) {
    const Res r = run<OP, false>(W, blocks, n, d_t, d_sink);
    printf(" %.2f (%.3f ns) |", r.cyc_memtime, r.ns_wall);
    if (W == 4) ns1 = r.ns_wall;
  }
  const Res d = run<OP, true>(1, blocks, n, d_t, d_sink);
  printf(" %.2f (%.3f ns) |\n", d.cyc_memtime, d.ns_wall);
  (void)ns1;
}

int main() {
  hipDeviceProp_t pr; (void)hipGetDeviceProperties(&pr, 0);
  const int blocks = pr.multiProcessorCount;  // one workgroup per CU
  const int n = 1024;
  uint64_t* d_t; int* d_sink;
  (void)hipMalloc(&d_t, (size_t)blocks * 32 * 4 * 8); (void)hipMalloc(&d_sink, (size_t)blocks * 2048 * 4);
  printf("# r6: integer VALU issue on %s (%s), %d CUs, clockRate %d kHz\n\n", pr.name, pr.gcnArchName, blocks, pr.clockRate);
  printf("`scripts/micro/valu_issue.hip`: one workgroup per CU of 4 W waves (W per SIMD), every wave %d x 32 instructions of one kind on 8 independent registers;\n", n);
  printf("figure = s_memtime ticks (and wall nanoseconds from s_memrealtime) per wave64 instruction per SIMD = span of the workgroup / (n x 32 x W), median over the CUs, best of 3.\n");
  printf("`dep` = one wave per SIMD, every instruction dependent on the one before (latency).\n\n");
  printf("| instruction | W=1 | W=2 | W=4 | W=8 | dep, W=1 |\n|---|---|---|---|---|---|\n");
  row<FMA_F32>(d_t, d_sink, blocks, n);
  row<MAX_I32>(d_t, d_sink, blocks, n);
  row<MAX3_I32>(d_t, d_sink, blocks, n);
  row<ADD_U32>(d_t, d_sink, blocks, n);
  row<ADD3_U32>(d_t, d_sink, blocks, n);
  row<LSHL_ADD>(d_t, d_sink, blocks, n);
  row<MIN_U32>(d_t, d_sink, blocks, n);
  row<CNDMASK>(d_t, d_sink, blocks, n);
  row<CMP_GT_I32>(d_t, d_sink, blocks, n);
  row<ALIGNBIT>(d_t, d_sink, blocks, n);
  row<XOR_B32>(d_t, d_sink, blocks, n);
  row<FFBL>(d_t, d_sink, blocks, n);
  row<BFE_U32>(d_t, d_sink, blocks, n);
  row<PERM_B32>(d_t, d_sink, blocks, n);
  row<MAD_U32_U24>(d_t, d_sink, blocks, n);
  row<PK_MAX_I16>(d_t, d_sink, blocks, n);
  row<PK_ADD_U16>(d_t, d_sink, blocks, n);
  row<CNDMASK_VCC_SET>(d_t, d_sink, blocks, n);
  row<CNDMASK_E64>(d_t, d_sink, blocks, n);
  row<CNDMASK_CONST>(d_t, d_sink, blocks, n);
  row<AND_B32>(d_t, d_sink, blocks, n);
  row<OR_B32>(d_t, d_sink, blocks, n);
  row<LSHLREV>(d_t, d_sink, blocks, n);
  row<LSHRREV>(d_t, d_sink, blocks, n);
  row<SUB_U32>(d_t, d_sink, blocks, n);
  row<MOV_B32>(d_t, d_sink, blocks, n);
  row<MAX_U32>(d_t, d_sink, blocks, n);
  row<MED3_I32>(d_t, d_sink, blocks, n);
  row<ADD_I32_CLAMP>(d_t, d_sink, blocks, n);
  row<SUBREV_CO>(d_t, d_sink, blocks, n);
  row<PAIR_VCC>(d_t, d_sink, blocks, n);
  row<PAIR_SGPR>(d_t, d_sink, blocks, n);
  row<PAIR_VCC_NONOP>(d_t, d_sink, blocks, n);
  row<CMP_E64>(d_t, d_sink, blocks, n);
  row<READLANE>(d_t, d_sink, blocks, n);
  row<MOV_DPP_ROWSHR>(d_t, d_sink, blocks, n);
  row<MOV_DPP_WAVESHR>(d_t, d_sink, blocks, n);
  row<MAX_DPP_ROWSHR>(d_t, d_sink, blocks, n);
  return 0;
}
