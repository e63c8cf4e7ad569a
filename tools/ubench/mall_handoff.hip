// Two measurements that decide the bf16 MLP step's next design:
//   A. Infinity Cache (256 MiB, memory side): bandwidth of reading a buffer that was just written / just read, by size, and what
//      survives a 2-GiB stream in between -- can a row chunk's saved state live there between forward, backward and weight gradient?
//   B. inter-CU hand-off: 128 producer workgroups hand 128-KiB tiles (one bf16 dz tile of 256 rows x 256 features) to 128 consumer
//      workgroups through L2 / the fabric with flags, a ring of 4 slots per pair: aggregate TB/s and the consumers' polling share
//      -- what a layer-pipelined backward (dz hopping from CU group to CU group, never written to HBM as a tensor) would ride on.
// Build: hipcc --offload-arch=gfx950 -O3 -o ubin/mall_handoff tools/ubench/mall_handoff.hip ; run: timeout 120 ubin/mall_handoff
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void write_k(f4* __restrict__ p, int64_t n, float v) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) p[i] = f4{v, v, v, v};
}
template <bool NT>
__global__ __launch_bounds__(256) void read_k(const f4* __restrict__ p, int64_t n, float* sink) {
  f4 acc = {0, 0, 0, 0};
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
    acc += NT ? __builtin_nontemporal_load(p + i) : p[i];
  if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) sink[0] = 1.f;
}

static float ms_of(hipEvent_t a, hipEvent_t b) { float m; CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&m, a, b)); return m; }

static void part_a() {
  const int64_t big = (int64_t)2 << 30;
  f4 *buf, *stream;
  float* sink;
  CK(hipMalloc(&buf, (size_t)4 << 30)); CK(hipMalloc(&stream, big)); CK(hipMalloc(&sink, 64));
  hipEvent_t e0, e1, e2;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&e2));
  const int grid = 256 * 8;
  hipLaunchKernelGGL(write_k, dim3(grid), dim3(256), 0, 0, stream, big / 16, 1.f);
  printf("A. Infinity Cache: GB/s by buffer size (median of 7; write = plain 16-B stores, read = plain 16-B loads)\n");
  printf("%8s %10s %12s %12s %22s %22s\n", "MiB", "write", "read-a-write", "read-a-read", "read after 2GiB stream", "... after nt stream");
  const int sizes[] = {16, 32, 64, 96, 128, 192, 256, 384, 512, 1024, 4096};
  for (int mb : sizes) {
    const int64_t n = (int64_t)mb * (1 << 20) / 16;
    std::vector<float> w, raw, rar, rs, rsn;
    for (int rep = 0; rep < 8; ++rep) {
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(write_k, dim3(grid), dim3(256), 0, 0, buf, n, (float)rep);
      CK(hipEventRecord(e1));
      hipLaunchKernelGGL(read_k<false>, dim3(grid), dim3(256), 0, 0, buf, n, sink);
      CK(hipEventRecord(e2));
      const float tw = ms_of(e0, e1), tr = ms_of(e1, e2);
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(read_k<false>, dim3(grid), dim3(256), 0, 0, buf, n, sink);
      CK(hipEventRecord(e1));
      const float trr = ms_of(e0, e1);
      float ts[2];
      for (int nt = 0; nt < 2; ++nt) {
        hipLaunchKernelGGL(write_k, dim3(grid), dim3(256), 0, 0, buf, n, (float)rep);
        if (nt) hipLaunchKernelGGL(read_k<true>, dim3(grid), dim3(256), 0, 0, stream, big / 16, sink);
        else hipLaunchKernelGGL(read_k<false>, dim3(grid), dim3(256), 0, 0, stream, big / 16, sink);
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(read_k<false>, dim3(grid), dim3(256), 0, 0, buf, n, sink);
        CK(hipEventRecord(e1));
        ts[nt] = ms_of(e0, e1);
      }
      if (rep) { w.push_back(tw); raw.push_back(tr); rar.push_back(trr); rs.push_back(ts[0]); rsn.push_back(ts[1]); }
    }
    auto med = [&](std::vector<float>& v) { std::sort(v.begin(), v.end()); return (double)mb * 1.048576 / v[v.size() / 2]; };
    printf("%8d %10.0f %12.0f %12.0f %22.0f %22.0f\n", mb, med(w), med(raw), med(rar), med(rs), med(rsn));
  }
  CK(hipFree(buf)); CK(hipFree(stream)); CK(hipFree(sink));
}

// ---- B. hand-off ------------------------------------------------------------------------------------------------------
constexpr int kSlotBytes = 128 << 10, kRing = 4, kSpinMax = 1 << 22;
struct HArgs {
  char* slots;            // [pairs][kRing][kSlotBytes]
  unsigned* ready;        // [pairs] tiles published (monotonic)
  unsigned* freed;        // [pairs] tiles consumed (monotonic)
  unsigned long long* poll_cycles;   // [pairs] consumer cycles spent polling
  unsigned long long* total_cycles;  // [pairs]
  unsigned* fail;
  float* sink;
  int pairs, tiles, shift, mode;     // mode 0: plain stores + agent release fence | acquire + plain loads; 1: sc0 sc1 stores + vmcnt(0) | sc1 loads
};

__device__ __forceinline__ unsigned ld_relaxed(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_relaxed(unsigned* p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__global__ __launch_bounds__(256) void handoff_k(HArgs a) {
  const int b = blockIdx.x, tid = threadIdx.x;
  const bool producer = b < a.pairs;
  const int pair = producer ? b : (b - a.pairs + a.shift) % a.pairs;     // shift 0: consumer b+pairs serves producer b (same XCD when pairs % 8 == 0)
  char* ring = a.slots + (size_t)pair * kRing * kSlotBytes;
  __shared__ unsigned ok;
  const unsigned long long t_begin = __builtin_readcyclecounter();
  unsigned long long polled = 0;
  f4 acc = {0, 0, 0, 0};
  for (int t = 0; t < a.tiles; ++t) {
    f4* slot = reinterpret_cast<f4*>(ring + (size_t)(t % kRing) * kSlotBytes);
    if (producer) {
      if (tid == 0) {      // ring full: wait for the consumer to free slot t % kRing
        int spin = 0;
        unsigned good = ld_relaxed(a.fail) == 0;
        while (good && t >= kRing && ld_relaxed(a.freed + pair) + kRing <= (unsigned)t) { __builtin_amdgcn_s_sleep(2); if (++spin > kSpinMax || ld_relaxed(a.fail)) { atomicAdd(a.fail, 1u); good = 0; } }
        ok = good;
      }
      __syncthreads();
      if (!ok) break;
      const f4 v = {(float)t, (float)pair, (float)tid, 1.f};
#pragma unroll 8
      for (int i = tid; i < kSlotBytes / 16; i += 256) {
        if (a.mode == 1) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(slot + i), "v"(v) : "memory");      // write-through
        else slot[i] = v;
      }
      if (a.mode == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // every wave drains its own write-through stores
      __syncthreads();
      if (tid == 0) {
        if (a.mode == 0) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        st_relaxed(a.ready + pair, (unsigned)t + 1);
      }
    } else {
      if (tid == 0) {
        const unsigned long long p0 = __builtin_readcyclecounter();
        int spin = 0;
        unsigned good = ld_relaxed(a.fail) == 0;
        while (good && ld_relaxed(a.ready + pair) <= (unsigned)t) { __builtin_amdgcn_s_sleep(2); if (++spin > kSpinMax || ld_relaxed(a.fail)) { atomicAdd(a.fail, 1u); good = 0; } }
        polled += __builtin_readcyclecounter() - p0;
        if (a.mode == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        ok = good;
      }
      __syncthreads();
      if (!ok) break;
      f4 v[8];
      for (int i0 = tid; i0 < kSlotBytes / 16; i0 += 256 * 8) {
        if (a.mode == 1) {      // sc1 loads bypass this CU's L1 (the producer stored write-through): no acquire needed
#pragma unroll
          for (int k = 0; k < 8; ++k) asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v[k]) : "v"(slot + i0 + k * 256) : "memory");
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
#pragma unroll
          for (int k = 0; k < 8; ++k) v[k] = slot[i0 + k * 256];
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) acc += v[k];
      }
      __syncthreads();
      if (tid == 0) st_relaxed(a.freed + pair, (unsigned)t + 1);
    }
  }
  if (tid == 0) {
    a.total_cycles[b] = __builtin_readcyclecounter() - t_begin;
    if (!producer) a.poll_cycles[pair] = polled;
  }
  if (acc[0] + acc[1] + acc[2] + acc[3] == -1.f) a.sink[0] = 1.f;
  // every consumer checks the LAST tile's content word for word (stale data would show here)
  if (!producer && a.tiles > 0) {
    const f4* slot = reinterpret_cast<const f4*>(ring + (size_t)((a.tiles - 1) % kRing) * kSlotBytes);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    for (int i = tid; i < kSlotBytes / 16; i += 256) {
      const f4 v = slot[i];
      if (v[0] != (float)(a.tiles - 1) || v[1] != (float)pair || v[2] != (float)(i % 256)) { atomicAdd(a.fail + 1, 1u); break; }
    }
  }
}

static void part_b() {
  printf("\nB. hand-off of 128-KiB tiles, producer workgroup -> consumer workgroup (1 per CU, 256 threads), ring of %d slots per pair\n", kRing);
  printf("%6s %22s %8s %10s %14s %14s %8s\n", "pairs", "placement", "tiles", "ms", "aggregate GB/s", "GB/s per pair", "polling");
  printf("%6s %22s %8s %10s %14s %14s %8s\n", "", "(first block: plain stores + agent release fence + flag | acquire fence + plain loads;", "", "", "", "", "");
  printf("%6s %22s\n", "", " second block: sc0 sc1 write-through stores + vmcnt(0) + flag | sc1 loads, no fence)");
  for (int mode : {0, 1})
  for (int pairs : {32, 64, 128}) {
    for (int shift : {0, 1}) {
      HArgs a;
      a.pairs = pairs; a.tiles = 400; a.shift = shift; a.mode = mode;
      CK(hipMalloc(&a.slots, (size_t)pairs * kRing * kSlotBytes));
      CK(hipMalloc(&a.ready, pairs * 4)); CK(hipMalloc(&a.freed, pairs * 4)); CK(hipMalloc(&a.fail, 8)); CK(hipMalloc(&a.sink, 64));
      CK(hipMalloc(&a.poll_cycles, pairs * 8)); CK(hipMalloc(&a.total_cycles, 2 * pairs * 8));
      hipEvent_t e0, e1;
      CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
      float best = 1e30f;
      double poll_share = 0;
      unsigned fails[2] = {0, 0};
      for (int rep = 0; rep < 3; ++rep) {
        CK(hipMemset(a.ready, 0, pairs * 4)); CK(hipMemset(a.freed, 0, pairs * 4)); CK(hipMemset(a.fail, 0, 8));
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(handoff_k, dim3(2 * pairs), dim3(256), 0, 0, a);
        CK(hipEventRecord(e1));
        const float ms = ms_of(e0, e1);
        std::vector<unsigned long long> pc(pairs), tc(2 * pairs);
        CK(hipMemcpy(pc.data(), a.poll_cycles, pairs * 8, hipMemcpyDeviceToHost));
        CK(hipMemcpy(tc.data(), a.total_cycles, 2 * pairs * 8, hipMemcpyDeviceToHost));
        CK(hipMemcpy(fails, a.fail, 8, hipMemcpyDeviceToHost));
        if (ms < best) {
          best = ms;
          double p = 0, t = 0;
          for (int i = 0; i < pairs; ++i) { p += (double)pc[i]; t += (double)tc[pairs + i]; }
          poll_share = p / t;
        }
      }
      const double gb = (double)pairs * a.tiles * kSlotBytes / 1e9;
      printf("%6d %22s %8d %10.3f %14.0f %14.1f %7.0f%%   %s\n", pairs, shift ? "consumer = next pair" : "same block index % 8", a.tiles, best,
             gb / (best * 1e-3), gb / (best * 1e-3) / pairs, 100 * poll_share, fails[0] || fails[1] ? "FAIL (timeouts / stale words)" : "data verified");
      CK(hipFree(a.slots)); CK(hipFree(a.ready)); CK(hipFree(a.freed)); CK(hipFree(a.fail)); CK(hipFree(a.sink)); CK(hipFree(a.poll_cycles)); CK(hipFree(a.total_cycles));
    }
  }
}

int main(int argc, char** argv) {
  hipDeviceProp_t p;
  CK(hipGetDeviceProperties(&p, 0));
  printf("%s, %d CUs, L2 %d MiB\n", p.name, p.multiProcessorCount, p.l2CacheSize >> 20);
  if (argc < 2 || argv[1][0] != 'b') part_a();
  if (argc < 2 || argv[1][0] != 'a') part_b();
  return 0;
}
