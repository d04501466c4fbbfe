// What bounds the 64-RHS stage kernel (bfStageKernelC128Mfma)?  Probes on the bare machine, each ~1 s so DVFS settles:
//   rate:  v_mfma_f64_16x16x4_f64 with 16 accumulators per wave at 2 (or 1) waves per SIMD,
//          operands constant vs random (data toggling = power), with and without the kernel's operand
//          traffic (2 KiB of A per k-step streamed from HBM with nt loads, 4 KiB of X per k-step from a
//          2 MB (L2) or 64 MB (MALL) region) requested one k-step ahead through buffer loads with SGPR offsets (no VALU at all
//          in the loop) -- the speed of light of the kernel's structure on this chip;
//   neg:   does blgp bit 0 of the f64 MFMA negate A (so that -Ai*Bi needs no v_xor)?
//   oob:   is the SGPR offset of a raw buffer load part of its range check?
//   hipcc -O3 --offload-arch=gfx950 tools/mfma_probe.hip -o /tmp/mfma_probe && /tmp/mfma_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef double bf_d4 __attribute__((ext_vector_type(4)));
typedef unsigned int u4 __attribute__((ext_vector_type(4)));
union Frag { u4 u; double d[2]; };

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ double rnd(uint32_t &s) {      // uniform in [-1, 1), full mantissa
  s = s * 1664525u + 1013904223u; uint32_t hi = s;
  s = s * 1664525u + 1013904223u; uint32_t lo = s;
  uint64_t bits = ((uint64_t)0x3ff << 52) | ((uint64_t)(hi & 0xfffff) << 32) | lo;
  return (__longlong_as_double((long long)bits) - 1.5) * 2.0;
}

// MODE bit 0: random operands; bit 1: X fragments loaded (L2-sized region); bit 2: A fragments loaded (HBM stream)
template <int MODE, int WPS>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WPS, WPS))) void probe(double *out, double2 const *A, double2 const *X,
                                                                                            uint32_t aWindow, uint32_t xWindow, int iters) {
  bf_d4 accr[2][4], acci[2][4];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int t = 0; t < 4; ++t) { accr[m][t] = (bf_d4){0, 0, 0, 0}; acci[m][t] = (bf_d4){0, 0, 0, 0}; }
  uint32_t seed = blockIdx.x * 64 + threadIdx.x + 12345u;
  Frag a[2][2], x[2][4];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
#pragma unroll
    for (int m = 0; m < 2; ++m) { a[s][m].d[0] = (MODE & 1) ? rnd(seed) : 1e-3; a[s][m].d[1] = (MODE & 1) ? rnd(seed) : 2e-3; }
#pragma unroll
    for (int t = 0; t < 4; ++t) { x[s][t].d[0] = (MODE & 1) ? rnd(seed) : 1e-3; x[s][t].d[1] = (MODE & 1) ? rnd(seed) : 3e-3; }
  }
  uint32_t const wave = blockIdx.x;
  // every wave streams its own window of A; groups of 8 consecutive waves walk one window of X together
  __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void *)((char const *)A + (uint64_t)wave * aWindow), 0, aWindow, 0x00020000);
  __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void *)X, 0, xWindow, 0x00020000);
  uint32_t const va = threadIdx.x * 16u, vx = threadIdx.x * 16u;
  uint32_t sa = 0, sx = ((wave >> 3) * 4096u * 97u) & (xWindow - 1);
  for (int it = 0; it < iters; it += 2) {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      // request the fragments of the NEXT k-step into set s ^ 1 ...
      if (MODE & 4) {
        sa = (sa + 2048u) & (aWindow - 1);
#pragma unroll
        for (int m = 0; m < 2; ++m) a[s ^ 1][m].u = __builtin_amdgcn_raw_buffer_load_b128(ra, va, sa + 1024u * m, 2);
      }
      if (MODE & 2) {
        sx = (sx + 4096u) & (xWindow - 1);
#pragma unroll
        for (int t = 0; t < 4; ++t) x[s ^ 1][t].u = __builtin_amdgcn_raw_buffer_load_b128(rx, vx, sx + 1024u * t, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      // ... and contract set s: a complex multiply-accumulate is 4 real MFMAs, -Ai*Bi through the NEG bit
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int m = 0; m < 2; ++m) {
          accr[m][t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[s][m].d[0], x[s][t].d[0], accr[m][t], 0, 0, 0);
          acci[m][t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[s][m].d[0], x[s][t].d[1], acci[m][t], 0, 0, 0);
        }
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int m = 0; m < 2; ++m) {
          accr[m][t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[s][m].d[1], x[s][t].d[1], accr[m][t], 0, 0, 1);
          acci[m][t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[s][m].d[1], x[s][t].d[0], acci[m][t], 0, 0, 0);
        }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  double sum = 0;
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int v = 0; v < 4; ++v) sum += accr[m][t][v] + acci[m][t][v];
  out[blockIdx.x * 64 + threadIdx.x] = sum;
}

// The same traffic with ONE set of fragment registers: a fragment is requested again right after the last MFMA that reads
// it (m-outer order: A[0] after 16 MFMAs, X[t] inside the second half, A[1] at the end), half a k-step ahead of its next
// use.  128 accumulator + 24 fragment registers leave room for WPS = 3 waves per SIMD (<= 168 VGPRs).  The loads are asm
// statements and the waits are placed by hand: hipcc's own wait insertion treats every load that is pending at a loop
// header as one lump (s_waitcnt vmcnt(0) at the top of the loop: the fragment requested last would be waited for first).
typedef int bf_i4 __attribute__((ext_vector_type(4)));
template <int NT_FLAG> __device__ __forceinline__ void asmLoad(Frag &f, uint32_t voff, bf_i4 rsrc, uint32_t soff) {
  if (NT_FLAG) asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen nt" : "=v"(f.u) : "v"(voff), "s"(rsrc), "s"(soff));
  else asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(f.u) : "v"(voff), "s"(rsrc), "s"(soff));
}
// a raw buffer descriptor in SGPRs: base (48 bits), stride 0, num_records in bytes, gfx9 data format word
__device__ __forceinline__ bf_i4 makeRsrc(void const *base, uint32_t bytes) {
  uint64_t const b = (uint64_t)base;
  bf_i4 r;
  r.x = __builtin_amdgcn_readfirstlane((int)(uint32_t)b);
  r.y = __builtin_amdgcn_readfirstlane((int)((uint32_t)(b >> 32) & 0xffffu));
  r.z = __builtin_amdgcn_readfirstlane((int)bytes);
  r.w = 0x00020000;
  return r;
}
template <int N> __device__ __forceinline__ void asmWait2(Frag &f, Frag &g) { asm volatile("s_waitcnt vmcnt(%2)" : "+v"(f.u), "+v"(g.u) : "n"(N)); }
template <int N> __device__ __forceinline__ void asmWait1(Frag &f) { asm volatile("s_waitcnt vmcnt(%1)" : "+v"(f.u) : "n"(N)); }

template <int WPS>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WPS, WPS))) void probeSingle(double *out, double2 const *A, double2 const *X,
                                                                                                 uint32_t aWindow, uint32_t xWindow, int iters) {
  bf_d4 accr[2][4], acci[2][4];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int t = 0; t < 4; ++t) { accr[m][t] = (bf_d4){0, 0, 0, 0}; acci[m][t] = (bf_d4){0, 0, 0, 0}; }
  uint32_t const wave = blockIdx.x;
  bf_i4 const ra = makeRsrc((char const *)A + (uint64_t)wave * aWindow, aWindow), rx = makeRsrc(X, xWindow);
  uint32_t const va = threadIdx.x * 16u, vx = threadIdx.x * 16u;
  uint32_t sa = 0, sx = ((wave >> 3) * 4096u * 97u) & (xWindow - 1);
  Frag a[2], x[4];
  asmLoad<1>(a[0], va, ra, sa);
#pragma unroll
  for (int t = 0; t < 4; ++t) asmLoad<0>(x[t], vx, rx, sx + 1024u * t);
  asmLoad<1>(a[1], va, ra, sa + 1024u);
  for (int it = 0; it < iters; ++it) {
    sa = (sa + 2048u) & (aWindow - 1);
    sx = (sx + 4096u) & (xWindow - 1);
#pragma unroll
    for (int m = 0; m < 2; ++m) {
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        // outstanding at the top of a step, in issue order: A0', X0' .. X3', A1'
        if (m == 0) { if (t == 0) asmWait2<4>(a[0], x[0]); else if (t == 1) asmWait1<3>(x[1]); else if (t == 2) asmWait1<2>(x[2]); else asmWait1<1>(x[3]); }
        else if (t == 0) asmWait1<1>(a[1]);          // A1', then the A0'' just requested
        accr[m][t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[m].d[0], x[t].d[0], accr[m][t], 0, 0, 0);
        acci[m][t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[m].d[0], x[t].d[1], acci[m][t], 0, 0, 0);
        accr[m][t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[m].d[1], x[t].d[1], accr[m][t], 0, 0, 1);
        acci[m][t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[m].d[1], x[t].d[0], acci[m][t], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (m == 1) { asmLoad<0>(x[t], vx, rx, sx + 1024u * t); __builtin_amdgcn_sched_barrier(0); }
      }
      asmLoad<1>(a[m], va, ra, sa + 1024u * m);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(a[0].u), "+v"(a[1].u), "+v"(x[0].u), "+v"(x[1].u), "+v"(x[2].u), "+v"(x[3].u));
  double sum = a[0].d[0] + a[1].d[0] + x[0].d[0] + x[1].d[0] + x[2].d[0] + x[3].d[0];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int v = 0; v < 4; ++v) sum += accr[m][t][v] + acci[m][t][v];
  out[blockIdx.x * 64 + threadIdx.x] = sum;
}

template <int WPS> static void runSingle(char const *name, double *out, double2 const *A, double2 const *X, uint32_t aWindow, uint32_t xWindow, int iters) {
  int const grid = 256 * 4 * WPS;
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  probeSingle<WPS><<<grid, 64>>>(out, A, X, aWindow, xWindow, 2000);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  probeSingle<WPS><<<grid, 64>>>(out, A, X, aWindow, xWindow, iters);
  CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
  double flops = (double)grid * iters * 32 * 2048.0;
  printf("\"%s\": {\"tflops\": %.2f, \"frac_of_78.6\": %.3f, \"ms\": %.1f, \"a_stream_tbs\": %.2f, \"x_l2_tbs\": %.2f, \"waves_per_simd\": %d},\n", name, flops / ms / 1e9,
         flops / ms / 1e9 / 78.6, ms, (double)grid * iters * 2048.0 / ms / 1e9, (double)grid * iters * 4096.0 / ms / 1e9, WPS);
  fflush(stdout);
}

__global__ void negProbe(double *out) {
  double a = 1.0 + threadIdx.x, b = 2.0 + (threadIdx.x & 15);
  bf_d4 z = (bf_d4){0, 0, 0, 0};
  bf_d4 p = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, z, 0, 0, 0);
  bf_d4 na = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, z, 0, 0, 1);
  bf_d4 nb = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, z, 0, 0, 2);
  bf_d4 c1 = (bf_d4){1, 1, 1, 1};
  bf_d4 nc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 4);
  out[threadIdx.x * 4 + 0] = p[0]; out[threadIdx.x * 4 + 1] = na[0]; out[threadIdx.x * 4 + 2] = nb[0]; out[threadIdx.x * 4 + 3] = nc[0];
}

__global__ void oobProbe(double2 const *p, double2 *o, uint32_t numRecords, uint32_t soff) {
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void *)p, 0, numRecords, 0x00020000);
  Frag f; f.u = __builtin_amdgcn_raw_buffer_load_b128(r, threadIdx.x * 16u, soff, 0);
  o[threadIdx.x] = make_double2(f.d[0], f.d[1]);
}

template <int MODE, int WPS> static void run(char const *name, double *out, double2 const *A, double2 const *X, uint32_t aWindow, uint32_t xWindow, int iters) {
  int const grid = 256 * 4 * WPS;
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  probe<MODE, WPS><<<grid, 64>>>(out, A, X, aWindow, xWindow, 2000);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  probe<MODE, WPS><<<grid, 64>>>(out, A, X, aWindow, xWindow, iters);
  CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
  double flops = (double)grid * iters * 32 * 2048.0;
  double hbm = (MODE & 4) ? (double)grid * iters * 2048.0 / ms / 1e9 : 0.0, l2 = (MODE & 2) ? (double)grid * iters * 4096.0 / ms / 1e9 : 0.0;
  printf("\"%s\": {\"tflops\": %.2f, \"frac_of_78.6\": %.3f, \"ms\": %.1f, \"a_stream_tbs\": %.2f, \"x_l2_tbs\": %.2f, \"waves_per_simd\": %d},\n", name, flops / ms / 1e9,
         flops / ms / 1e9 / 78.6, ms, hbm, l2, WPS);
  fflush(stdout);
}

int main(int argc, char **argv) {
  int const iters = argc > 1 ? atoi(argv[1]) : 500000;
  uint32_t const aWindow = 8u << 20, xWindow = 64u << 20;
  double *out; double2 *A, *X;
  CHECK(hipMalloc(&out, 256 * 4 * 3 * 64 * 8));
  CHECK(hipMalloc(&A, (size_t)aWindow * 3072));
  CHECK(hipMalloc(&X, xWindow));
  {   // finite random contents (NaN-free) for the loaded operands
    std::vector<double> h((size_t)(64u << 20) / 8);
    uint64_t s = 88172645463325252ull;
    for (auto &v : h) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; v = (double)(int64_t)s / 9.3e18; }
    for (size_t off = 0; off < (size_t)aWindow * 3072; off += h.size() * 8) CHECK(hipMemcpy((char *)A + off, h.data(), h.size() * 8, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(X, h.data(), xWindow, hipMemcpyHostToDevice));
  }
  printf("{\n");
  run<0, 2>("const_operands", out, A, X, aWindow, xWindow, iters);
  run<1, 2>("random_operands", out, A, X, aWindow, xWindow, iters);
  run<1, 1>("random_operands_1wave", out, A, X, aWindow, xWindow, iters);
  run<3, 2>("random_x_from_2mb", out, A, X, aWindow, 2u << 20, iters);
  run<3, 2>("random_x_from_64mb", out, A, X, aWindow, xWindow, iters);
  run<5, 2>("random_a_from_hbm", out, A, X, aWindow, xWindow, iters);
  run<7, 2>("a_from_hbm_x_from_2mb", out, A, X, aWindow, 2u << 20, iters);
  run<7, 2>("a_from_hbm_x_from_64mb", out, A, X, aWindow, xWindow, iters);
  run<7, 1>("a_from_hbm_x_from_2mb_1wave", out, A, X, aWindow, 2u << 20, iters);
  runSingle<2>("single_set_a_hbm_x_2mb_2waves", out, A, X, aWindow, 2u << 20, iters);
  runSingle<3>("single_set_a_hbm_x_2mb_3waves", out, A, X, aWindow, 2u << 20, iters * 2 / 3);
  runSingle<3>("single_set_a_hbm_x_64mb_3waves", out, A, X, aWindow, xWindow, iters * 2 / 3);
  // NEG bits of the f64 MFMA
  {
    double *d; CHECK(hipMalloc(&d, 64 * 4 * 8)); double h[256];
    negProbe<<<1, 64>>>(d); CHECK(hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost));
    int na = 1, nb = 1, nc = 1;
    for (int l = 0; l < 64; ++l) { na &= h[4 * l + 1] == -h[4 * l]; nb &= h[4 * l + 2] == -h[4 * l]; nc &= h[4 * l + 3] == h[4 * l] - 1.0; }
    printf("\"mfma_f64_blgp\": {\"bit0_negates_a\": %d, \"bit1_negates_b\": %d, \"bit2_negates_c\": %d, \"sample\": [%.1f, %.1f, %.1f, %.1f]},\n", na, nb, nc, h[0], h[1], h[2], h[3]);
  }
  // range check of raw buffer loads: 1024-byte buffer inside a 4 KiB allocation of ones
  {
    double2 *p, *o; CHECK(hipMalloc(&p, 4096)); CHECK(hipMalloc(&o, 1024)); double2 hp[256], ho[64];
    for (int i = 0; i < 256; ++i) hp[i] = make_double2(i + 1.0, 0.0);
    CHECK(hipMemcpy(p, hp, sizeof hp, hipMemcpyHostToDevice));
    oobProbe<<<1, 64>>>(p, o, 1024, 512); CHECK(hipMemcpy(ho, o, sizeof ho, hipMemcpyDeviceToHost));
    // lane l reads byte offset 16 l + 512: in range of the 1024-byte buffer iff l < 32 when soffset counts
    printf("\"raw_buffer_soffset\": {\"lane31\": %.1f, \"lane32\": %.1f, \"lane63\": %.1f, \"soffset_is_range_checked\": %d}\n", ho[31].x, ho[32].x, ho[63].x, ho[32].x == 0.0);
  }
  printf("}\n");
  return 0;
}
