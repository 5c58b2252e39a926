// Micro-benchmark (diagnostic, not part of the library): cycles per instruction of the building blocks of the register-resident
// panel sweep on gfx950 -- v_fmac_f64 with a row_newbcast DPP operand against the plain instruction, the broadcast as a separate
// v_mov_b64_dpp, v_permlane swaps, and the latency of a dependent fp64 chain. One wavefront per SIMD (256 threads), s_memtime around
// unrolled blocks.   hipcc --offload-arch=gfx950 -O3 scripts/dpp_probe.hip -o /tmp/dpp_probe && /tmp/dpp_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP16(X) X X X X X X X X X X X X X X X X

__global__ void probe(unsigned long long* out, double* sink, double seed) {
  double x[16], a0 = 0, a1 = 0, a2 = 0, a3 = 0, n = seed * 1e-3;
  for (int i = 0; i < 16; ++i) x[i] = seed + threadIdx.x * 1e-3 + i;
  unsigned long long t[8];
  auto now = [] { return __builtin_amdgcn_s_memtime(); };
  // (0) 64 DPP fmac, dot style, four accumulators
  t[0] = now();
#pragma unroll
  for (int rep = 0; rep < 4; ++rep)
    asm volatile(
        "s_nop 1\n\t"
        "v_fmac_f64_dpp %0, %4, %4 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %1, %5, %5 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %2, %6, %6 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %3, %7, %7 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %0, %8, %8 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %1, %9, %9 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %2, %10, %10 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %3, %11, %11 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %0, %12, %12 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %1, %13, %13 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %2, %14, %14 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %3, %15, %15 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %0, %16, %16 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %1, %17, %17 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %2, %18, %18 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %3, %19, %19 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 0"
        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3)
        : "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]), "v"(x[6]), "v"(x[7]), "v"(x[8]), "v"(x[9]), "v"(x[10]),
          "v"(x[11]), "v"(x[12]), "v"(x[13]), "v"(x[14]), "v"(x[15]));
  t[1] = now();
  // (1) 64 plain fmac, four accumulators
#pragma unroll
  for (int rep = 0; rep < 4; ++rep)
    asm volatile(
        "v_fmac_f64 %0, %4, %4\n\tv_fmac_f64 %1, %5, %5\n\tv_fmac_f64 %2, %6, %6\n\tv_fmac_f64 %3, %7, %7\n\t"
        "v_fmac_f64 %0, %8, %8\n\tv_fmac_f64 %1, %9, %9\n\tv_fmac_f64 %2, %10, %10\n\tv_fmac_f64 %3, %11, %11\n\t"
        "v_fmac_f64 %0, %12, %12\n\tv_fmac_f64 %1, %13, %13\n\tv_fmac_f64 %2, %14, %14\n\tv_fmac_f64 %3, %15, %15\n\t"
        "v_fmac_f64 %0, %16, %16\n\tv_fmac_f64 %1, %17, %17\n\tv_fmac_f64 %2, %18, %18\n\tv_fmac_f64 %3, %19, %19"
        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3)
        : "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]), "v"(x[6]), "v"(x[7]), "v"(x[8]), "v"(x[9]), "v"(x[10]),
          "v"(x[11]), "v"(x[12]), "v"(x[13]), "v"(x[14]), "v"(x[15]));
  t[2] = now();
  // (2) 64 DPP fmac, update style (independent destinations)
#pragma unroll
  for (int rep = 0; rep < 4; ++rep)
    asm volatile(
        "s_nop 1\n\t"
        "v_fmac_f64_dpp %0, %0, %16 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %1, %1, %16 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %2, %2, %16 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %3, %3, %16 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %4, %4, %16 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %5, %5, %16 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %6, %6, %16 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %7, %7, %16 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %8, %8, %16 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %9, %9, %16 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %10, %10, %16 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %11, %11, %16 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %12, %12, %16 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %13, %13, %16 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %14, %14, %16 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %15, %15, %16 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 0"
        : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]), "+v"(x[8]), "+v"(x[9]),
          "+v"(x[10]), "+v"(x[11]), "+v"(x[12]), "+v"(x[13]), "+v"(x[14]), "+v"(x[15])
        : "v"(n));
  t[3] = now();
  // (3) 64 v_mov_b64_dpp broadcasts (independent)
  double y[16];
#pragma unroll
  for (int rep = 0; rep < 4; ++rep)
#pragma unroll
    for (int i = 0; i < 16; ++i)
      asm volatile("v_mov_b64_dpp %0, %1 row_newbcast:5 row_mask:0xf bank_mask:0xf" : "=v"(y[i]) : "v"(x[i]));
  t[4] = now();
  // (4) a chain of 64 dependent plain fma
  double c = a0;
#pragma unroll
  for (int i = 0; i < 64; ++i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(c) : "v"(n), "v"(seed));
  t[5] = now();
  // (5) a chain of 32 dependent DPP fmac
#pragma unroll
  for (int i = 0; i < 32; ++i) asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %0 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(c) : "v"(n));
  t[6] = now();
  // (6) 16 x (rsq + rcp) dependent
#pragma unroll
  for (int i = 0; i < 16; ++i) asm volatile("v_rsq_f64 %0, %0\n\tv_rcp_f64 %0, %0" : "+v"(c));
  t[7] = now();
  double s = a0 + a1 + a2 + a3 + c;
  for (int i = 0; i < 16; ++i) s += x[i] + y[i];
  sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0)
    for (int i = 0; i < 7; ++i) out[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 8 + i] = t[i + 1] - t[i];
}

int main() {
  unsigned long long* d;
  double* sink;
  hipMalloc(&d, 4 * 8 * 8 * sizeof(unsigned long long));
  hipMalloc(&sink, 4 * 256 * sizeof(double));
  for (int waves = 1; waves <= 2; ++waves) {
    for (int it = 0; it < 3; ++it) hipLaunchKernelGGL(probe, dim3(1), dim3(256 * waves), 0, 0, d, sink, 1.0 + it);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(64);
    hipMemcpy(h.data(), d, 64 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    const char* names[7] = {"64 dpp fmac (dot, 4 acc)", "64 plain fmac (4 acc)", "64 dpp fmac (update)", "64 v_mov_b64_dpp",
                            "64 dependent plain fma", "32 dependent dpp fmac (+s_nop 1)", "16 x (rsq + rcp) dependent"};
    printf("%d wavefront(s) per SIMD: cycles of wavefront 0\n", waves);
    for (int i = 0; i < 7; ++i) printf("  %-36s %6llu\n", names[i], h[i]);
  }
  return 0;
}
