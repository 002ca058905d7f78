// Ceiling experiment: the chain kernel's HBM access shape with no compute.
// variant 0: 16 x dwordx2 strided loads + 16 x dword stores per thread (what chain1d does)
// variant 1: 8 x dwordx4 loads (two adjacent samples) + 4 x dwordx4 stores
// variant 2: variant-0 loads + 4 x dwordx4 stores of 16 CONSECUTIVE words per thread (64-B lane stride)
// variant 3: 8 x dwordx4 loads, each half-wave on its own row (512 B per row) + 4 x dwordx4 contiguous stores
// variant 4: variant-0 loads + 4 x dwordx4 contiguous stores (1 KB per wave-instruction)
// LDS bytes per workgroup selectable to reproduce the occupancy (37 KB -> 4 WG/CU).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));

template <int VAR>
__global__ void __launch_bounds__(256) k(const float* __restrict__ in, unsigned* __restrict__ out, int spin) {
  extern __shared__ float lds[];
  const int tau = threadIdx.x;
  const size_t frame = blockIdx.x;
  float acc = 0.f;
  if (VAR == 0) {
    const f2* src = reinterpret_cast<const f2*>(in) + frame * 4096;
    f2 x[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) x[e] = src[e * 256 + tau];
#pragma unroll
    for (int e = 0; e < 16; ++e) acc += x[e].x * x[e].y;
    for (int i = 0; i < spin; ++i) acc = __fmaf_rn(acc, 1.0001f, 0.5f);
    lds[tau] = acc; __syncthreads(); acc = lds[tau ^ 1];
    unsigned* dst = out + frame * 4096;
#pragma unroll
    for (int j = 0; j < 16; ++j) dst[j * 256 + tau] = __float_as_uint(acc) + j;
  } else if (VAR == 2 || VAR == 4) {
    const f2* src = reinterpret_cast<const f2*>(in) + frame * 4096;
    f2 x[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) x[e] = src[e * 256 + tau];
#pragma unroll
    for (int e = 0; e < 16; ++e) acc += x[e].x * x[e].y;
    for (int i = 0; i < spin; ++i) acc = __fmaf_rn(acc, 1.0001f, 0.5f);
    lds[tau] = acc; __syncthreads(); acc = lds[tau ^ 1];
    u4* dst = reinterpret_cast<u4*>(out) + frame * 1024;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      u4 v = {__float_as_uint(acc), (unsigned)j, 1u, 2u};
      if (VAR == 2) dst[4 * tau + j] = v; else dst[j * 256 + tau] = v;
    }
  } else if (VAR == 3) {
    // lanes 0-31: row 2e, samples 2l, 2l+1 of the wave's 64 columns; lanes 32-63: row 2e+1
    const int lane = tau & 63, wave = tau >> 6;
    const f4* src = reinterpret_cast<const f4*>(in) + frame * 2048 + wave * 32 + (lane & 31) + (lane >> 5) * 128;
    f4 x[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) x[e] = src[e * 256];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc += x[e].x * x[e].y + x[e].z * x[e].w;
    for (int i = 0; i < spin; ++i) acc = __fmaf_rn(acc, 1.0001f, 0.5f);
    lds[tau] = acc; __syncthreads(); acc = lds[tau ^ 1];
    u4* dst = reinterpret_cast<u4*>(out) + frame * 1024;
#pragma unroll
    for (int j = 0; j < 4; ++j) { u4 v = {__float_as_uint(acc), (unsigned)j, 1u, 2u}; dst[j * 256 + tau] = v; }
  } else {
    const f4* src = reinterpret_cast<const f4*>(in) + frame * 2048;
    f4 x[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) x[e] = src[e * 256 + tau];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc += x[e].x * x[e].y + x[e].z * x[e].w;
    for (int i = 0; i < spin; ++i) acc = __fmaf_rn(acc, 1.0001f, 0.5f);
    lds[tau] = acc; __syncthreads(); acc = lds[tau ^ 1];
    u4* dst = reinterpret_cast<u4*>(out) + frame * 1024;
#pragma unroll
    for (int j = 0; j < 4; ++j) { u4 v = {__float_as_uint(acc), (unsigned)j, 1u, 2u}; dst[j * 256 + tau] = v; }
  }
}

int main(int argc, char** argv) {
  const int frames = 4096, sets = 4, reps = 40;
  std::vector<float*> in(sets); std::vector<unsigned*> out(sets);
  for (int s = 0; s < sets; ++s) { hipMalloc(&in[s], (size_t)frames * 4096 * 8); hipMalloc(&out[s], (size_t)frames * 4096 * 4); hipMemset(in[s], 1, (size_t)frames * 4096 * 8); }
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int var = 0; var < 5; ++var)
    for (int lds : {38000, 52000})
      for (int spin : {0, 2000}) {
        auto fn = var == 0 ? k<0> : var == 1 ? k<1> : var == 2 ? k<2> : var == 3 ? k<3> : k<4>;
        hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        for (int i = 0; i < 8; ++i) hipLaunchKernelGGL(fn, dim3(frames), dim3(256), lds, 0, in[i % sets], out[i % sets], spin);
        hipEventRecord(e0);
        for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(fn, dim3(frames), dim3(256), lds, 0, in[i % sets], out[i % sets], spin);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("var %d lds %5d spin %4d: %.1f us/launch  %.2f TB/s\n", var, lds, spin, ms / reps * 1e3, 201.3e6 / (ms / reps * 1e-3) / 1e12);
      }
  return 0;
}
