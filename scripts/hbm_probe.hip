// Streaming-bandwidth probe (hipcc -O3 --offload-arch=gfx950 scripts/hbm_probe.hip -o hbm_probe): what does the strip access
// pattern of the LQ kernels get from HBM without any compute? Numbers quoted in DESIGN.md section 6.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
constexpr int D = 256;
// pattern 1: the LQ strip layout (lane (r, g): row r, 16-column chunks 64 T + 16 g), read + scale + write back
__global__ __launch_bounds__(256, 2) void strip_rw(double* M, int nstrips, int inflight) {
  const int b = blockIdx.x, lane = threadIdx.x & 63, wv = threadIdx.x >> 6, r = lane & 15, g = lane >> 4;
  double* Mt = M + size_t(b) * D * D;
  for (int st = wv; st < nstrips; st += 4) {
    double* rowp = Mt + size_t(16 * st + r) * D + 16 * g;
    double X[64];
#pragma unroll
    for (int T4 = 0; T4 < 4; ++T4)
#pragma unroll
      for (int e = 0; e < 16; e += 2) {
        const double2 t = *reinterpret_cast<const double2*>(rowp + 64 * T4 + e);
        X[16 * T4 + e] = t.x; X[16 * T4 + e + 1] = t.y;
      }
    if (inflight) for (int z = 0; z < inflight; ++z) __builtin_amdgcn_s_sleep(127);
#pragma unroll
    for (int T4 = 0; T4 < 4; ++T4)
#pragma unroll
      for (int e = 0; e < 16; e += 2) {
        double2 t; t.x = X[16 * T4 + e] * 1.0000001; t.y = X[16 * T4 + e + 1] * 1.0000001;
        *reinterpret_cast<double2*>(rowp + 64 * T4 + e) = t;
      }
  }
}
// pattern 2: plain coalesced grid-stride read-modify-write
__global__ __launch_bounds__(256) void plain_rw(double2* M, size_t n) {
  for (size_t i = size_t(blockIdx.x) * 256 + threadIdx.x; i < n; i += size_t(gridDim.x) * 256) {
    double2 t = M[i]; t.x *= 1.0000001; t.y *= 1.0000001; M[i] = t;
  }
}
__global__ __launch_bounds__(256) void plain_r(const double2* M, size_t n, double* out) {
  double acc = 0;
  for (size_t i = size_t(blockIdx.x) * 256 + threadIdx.x; i < n; i += size_t(gridDim.x) * 256) { double2 t = M[i]; acc += t.x + t.y; }
  if (acc == 123.456) out[0] = acc;
}
int main() {
  const int B = 1024;
  const size_t n = size_t(B) * D * D;
  double* M; CK(hipMalloc(&M, n * 8)); CK(hipMemset(M, 0, n * 8));
  double* out; CK(hipMalloc(&out, 8));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto timeit = [&](const char* name, auto launch, double bytes) {
    launch(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0)); for (int i = 0; i < 10; ++i) launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 10;
    printf("%-44s %8.3f ms  %7.2f TB/s\n", name, ms, bytes / ms / 1e9);
  };
  timeit("strip layout r+w, 14 strips/traj", [&] { hipLaunchKernelGGL(strip_rw, dim3(B), dim3(256), 0, 0, M, 14, 0); }, 2.0 * B * 14 * 16 * D * 8);
  timeit("strip layout r+w, 16 strips/traj", [&] { hipLaunchKernelGGL(strip_rw, dim3(B), dim3(256), 0, 0, M, 16, 0); }, 2.0 * n * 8);
  timeit("strip layout r+w + sleep(64) between", [&] { hipLaunchKernelGGL(strip_rw, dim3(B), dim3(256), 0, 0, M, 16, 64); }, 2.0 * n * 8);
  timeit("plain coalesced r+w (4096 WGs)", [&] { hipLaunchKernelGGL(plain_rw, dim3(4096), dim3(256), 0, 0, (double2*)M, n / 2); }, 2.0 * n * 8);
  timeit("plain coalesced read (4096 WGs)", [&] { hipLaunchKernelGGL(plain_r, dim3(4096), dim3(256), 0, 0, (const double2*)M, n / 2, out); }, 1.0 * n * 8);
  return 0;
}
