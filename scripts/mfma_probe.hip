// fp64 matrix-core probe (hipcc -O3 --offload-arch=gfx950 scripts/mfma_probe.hip -o mfma_probe): what does v_mfma_f64_16x16x4_f64 sustain on
// this part with operands in registers and nothing else going on -- the practical ceiling under the 78.6 TFLOP/s the roofline uses.
// W wavefronts per SIMD, each with 16 independent accumulators (the dependency pattern of the 128 x 128 GEMM tile).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef double double4_t __attribute__((ext_vector_type(4)));
__device__ unsigned long long g_cycles[2];
template <int NACC>
__global__ __launch_bounds__(256) void mfma_loop(double* out, int iters) {
  double4_t acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = double4_t{0.0, 0.0, 0.0, 0.0};
  // distinct operand registers per MFMA of a group of four (the 4 x 4 tile of a GEMM reads 4 + 4 operands for 16 MFMAs)
  double a[4], b[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) { a[i] = 1.0 + 1e-9 * (threadIdx.x + i); b[i] = 1.0 - 1e-9 * (threadIdx.x + i); }
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), w0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i)  // (inline assembly: the compiler otherwise moves the accumulators between VGPRs and AGPRs every iteration)
      asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a[(i >> 2) & 3]), "v"(b[i & 3]));
#pragma unroll
    for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(a[i]), "+v"(b[i]));
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime(), w1 = wall_clock64();
  if (blockIdx.x == 0 && threadIdx.x == 0) { g_cycles[0] = c1 - c0; g_cycles[1] = w1 - w0; }
  double s = 0.0;
#pragma unroll
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[size_t(blockIdx.x) * 256 + threadIdx.x] = s;
}
template <int NACC>
static void run(int wgs_per_cu, int iters, double* out) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int grid = 256 * wgs_per_cu;
  hipLaunchKernelGGL(mfma_loop<NACC>, dim3(grid), dim3(256), 0, 0, out, iters / 10);
  float best = 1e30f;
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(mfma_loop<NACC>, dim3(grid), dim3(256), 0, 0, out, iters);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    best = ms < best ? ms : best;
  }
  const double flops = double(grid) * 4 * double(iters) * NACC * 2048.0;
  unsigned long long cyc[2];
  CK(hipMemcpyFromSymbol(cyc, HIP_SYMBOL(g_cycles), sizeof(cyc)));
  printf("accumulators %2d, waves per SIMD %d: %.3f ms, %.1f TFLOP/s | one wavefront: %.1f s_memtime ticks per MFMA, s_memtime / wall_clock64 = %.2f\n", NACC,
         wgs_per_cu, best, flops / best * 1e-9, double(cyc[0]) / (double(iters) * NACC), double(cyc[0]) / double(cyc[1]));
}
int main() {
  double* out;
  CK(hipMalloc(&out, sizeof(double) * 256 * 256 * 8));
  run<16>(1, 20000, out);
  run<16>(2, 20000, out);
  run<16>(4, 10000, out);
  run<4>(1, 40000, out);
  run<4>(2, 40000, out);
  run<1>(1, 40000, out);
  run<1>(4, 40000, out);
  // a long run: does the rate hold over 2 s (clocks under sustained fp64 matrix load)?
  run<16>(2, 2000000, out);
  return 0;
}
