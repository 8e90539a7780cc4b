// Premise check for the XCD-resident iteration (DESIGN.md "k_resident"): what does an exchange between the blocks of
// ONE XCD cost when it goes through that XCD's L2 (plain stores + L1-bypassing loads) instead of through the memory
// side (sc1 stores + sc1 loads, the placement-independent form), and is it ever stale?
//
//   part 1  placement census: XCC_ID of every block of an 8 x R grid, alone and with four grids on four streams.
//   part 2  ping-pong between two blocks (same XCD / different XCDs), one-way hop latency per store / load flavour.
//   part 3  reduce + broadcast among NB blocks of one XCD (what one phase of an optimiser iteration needs): every
//           block stores a partial, arrives on a counter, the last arriver reduces all partials and publishes a
//           48-float result + generation word, everybody reads it.  Every word is checked against its expected
//           value (a stale read is counted), blocks do uneven filler work between rounds and re-read the same lines
//           every round (L1-warm consumers), as MI355X_MICROARCH.md asks of a hand-off test.
// Build: hipcc --offload-arch=gfx950 -O3 xcd_exchange.hip -o xcd_exchange
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>

#define CK(x)                                                                  \
  do {                                                                         \
    hipError_t e = (x);                                                        \
    if (e != hipSuccess) {                                                     \
      printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); \
      return 1;                                                                \
    }                                                                          \
  } while (0)

__device__ __forceinline__ unsigned xcc_id() {
  unsigned v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return v & 0xf;
}
__device__ __forceinline__ unsigned long long now() { return __builtin_amdgcn_s_memrealtime(); }  // 100 MHz

// ---- access flavours ---------------------------------------------------------------------------------------------
enum { ST_PLAIN = 0, ST_SC1 = 1 };
enum { LD_PLAIN = 0, LD_SC1 = 1, LD_NT = 2, LD_ATOMIC_WG = 3, LD_SC0SC1 = 4 };
template <int F>
__device__ __forceinline__ void st32(unsigned* p, unsigned v) {
  if (F == ST_SC1)
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else
    asm volatile("global_store_dword %0, %1, off" ::"v"(p), "v"(v) : "memory");
}
template <int F>
__device__ __forceinline__ unsigned ld32(unsigned* p) {
  unsigned v;
  if (F == LD_SC1)
    asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  else if (F == LD_NT)
    asm volatile("global_load_dword %0, %1, off nt\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  else if (F == LD_SC0SC1)
    asm volatile("global_load_dword %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  else if (F == LD_ATOMIC_WG)
    v = __hip_atomic_fetch_or(p, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  else
    asm volatile("global_load_dword %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return v;
}
template <int F>
__device__ __forceinline__ double ld64(const double* p) {
  double v;
  if (F == LD_SC1)
    asm volatile("global_load_dwordx2 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  else if (F == LD_NT)
    asm volatile("global_load_dwordx2 %0, %1, off nt\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  else if (F == LD_SC0SC1)
    asm volatile("global_load_dwordx2 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  else
    asm volatile("global_load_dwordx2 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return v;
}
template <int F>
__device__ __forceinline__ void st64(double* p, double v) {
  if (F == ST_SC1)
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else
    asm volatile("global_store_dwordx2 %0, %1, off" ::"v"(p), "v"(v) : "memory");
}

// ---- part 1 ----------------------------------------------------------------------------------------------------
__global__ void k_census(unsigned* xcc, int hold_ticks) {
  if (threadIdx.x == 0) xcc[blockIdx.x] = xcc_id();
  const unsigned long long t0 = now();
  while (now() - t0 < (unsigned long long)hold_ticks) __builtin_amdgcn_s_sleep(8);
}

// ---- part 2: ping-pong.  Block A = blockIdx a, block B = blockIdx b; everybody else leaves. --------------------
template <int ST, int LD>
__global__ void k_pingpong(unsigned* flags, int a, int b, int rounds, unsigned long long* out, unsigned* xcc_out) {
  if ((int)blockIdx.x != a && (int)blockIdx.x != b) return;
  if (threadIdx.x != 0) return;
  const bool isA = (int)blockIdx.x == a;
  unsigned* mine = flags + (isA ? 0 : 64);    // separate cache lines
  unsigned* theirs = flags + (isA ? 64 : 0);
  xcc_out[isA ? 0 : 1] = xcc_id();
  const unsigned long long t0 = now();
  unsigned long long tmo = 0;
  for (int r = 1; r <= rounds && !tmo; r++) {
    if (isA) {
      st32<ST>(mine, (unsigned)r);
      const unsigned long long w0 = now();
      while (ld32<LD>(theirs) != (unsigned)r)
        if (now() - w0 > 2000000ull) {  // 20 ms: a stale flavour never sees the value
          tmo = 1;
          break;
        }
    } else {
      const unsigned long long w0 = now();
      while (ld32<LD>(theirs) != (unsigned)r)
        if (now() - w0 > 2000000ull) {
          tmo = 1;
          break;
        }
      st32<ST>(mine, (unsigned)r);
    }
  }
  if (isA) {
    out[0] = now() - t0;
    out[1] = tmo;
  }
}

// ---- part 3: reduce + broadcast among the NB blocks of each XCD's team ------------------------------------------
struct Team {
  double part[64][8];      // one 64-byte line per block
  float result[48];
  unsigned gen;            // generation of `result`
  unsigned pad0[15];
  unsigned arrive;         // monotonic arrival counter
  unsigned pad1[15];
  unsigned n_members;      // blocks that joined this team (self-placement)
  unsigned pad2[15];
};
struct Report {
  unsigned long long ticks, stale_part, stale_result, timeouts, rounds_done;
};

template <int ST, int LD, bool AGENT_ATOMIC>
__global__ __launch_bounds__(256) void k_team(Team* teams, int NB, int rounds, int filler, const float* junk, Report* rep,
                                              float* sink) {
  __shared__ int s_role, s_last;
  __shared__ unsigned s_team;
  // self-placement: the team is the XCD the block actually runs on
  if (threadIdx.x == 0) {
    const unsigned x = xcc_id();
    s_team = x;
    s_role = (int)atomicAdd(&teams[x].n_members, 1u);  // (device scope: only an index)
  }
  __syncthreads();
  Team* T = teams + s_team;
  const int role = s_role;
  if (role >= NB) return;  // surplus block of this XCD
  const int lane = threadIdx.x & 63;
  unsigned long long stale_p = 0, stale_r = 0, tmo = 0;
  float acc = 0.f;
  unsigned long long t0 = 0;
  int r = 1;
  for (; r <= rounds && !tmo; r++) {
    if (r == 2) t0 = now();  // (round 1 includes the team's start-up skew)
    // uneven filler work: block `role` streams (role % 4 + 1) * filler floats through its L1
    for (int q = threadIdx.x; q < (role % 4 + 1) * filler; q += 256) acc += junk[(size_t)role * 65536 + q];
    // partial of this round: 8 doubles, value encodes (round, role, component)
    if (threadIdx.x < 8) st64<ST>(&T->part[role][threadIdx.x], (double)(r * 1000 + role) + 0.125 * threadIdx.x);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
      unsigned done;
      if (AGENT_ATOMIC)
        done = __hip_atomic_fetch_add(&T->arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else
        done = __hip_atomic_fetch_add(&T->arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      s_last = (done == (unsigned)(r * NB - 1)) ? 1 : 0;
    }
    __syncthreads();
    if (s_last) {
      if (threadIdx.x < 64) {
        // lane l owns component l & 7 of blocks l >> 3, l >> 3 + 8, ...
        double s = 0;
        for (int b = lane >> 3; b < NB; b += 8) {
          const double v = ld64<LD>(&T->part[b][lane & 7]);
          const double want = (double)(r * 1000 + b) + 0.125 * (lane & 7);
          if (v != want) stale_p++;
          s += v;
        }
        // "result": 48 floats that encode the round
        if (lane < 48) {
          const float out = (float)(r * 64 + lane) + (float)(s * 0.0);
          if (ST == ST_SC1)
            __hip_atomic_store(&T->result[lane], out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          else
            asm volatile("global_store_dword %0, %1, off" ::"v"(&T->result[lane]), "v"(out) : "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) st32<ST>(&T->gen, (unsigned)r);
      }
    }
    // everybody (the publisher included) waits for the generation, then reads the result
    if (threadIdx.x < 64) {
      const unsigned long long w0 = now();
      unsigned g = 0;
      if (lane == 0) {
        while ((g = ld32<LD>(&T->gen)) < (unsigned)r) {
          if (now() - w0 > 3000000ull) {  // 30 ms
            tmo = 1;
            break;
          }
          __builtin_amdgcn_s_sleep(1);
        }
      }
      tmo = __shfl(tmo, 0);
      if (!tmo && lane < 48) {
        float v;
        float* p = &T->result[lane];
        if (LD == LD_SC1)
          asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
        else if (LD == LD_NT)
          asm volatile("global_load_dword %0, %1, off nt\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
        else if (LD == LD_SC0SC1)
          asm volatile("global_load_dword %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
        else
          asm volatile("global_load_dword %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
        // (a later round's result may already be there if this block is slow: only OLDER values are stale)
        if (v < (float)(r * 64 + lane)) stale_r++;
      }
    }
    __syncthreads();
  }
  if (threadIdx.x < 64) {
    for (int o = 32; o > 0; o >>= 1) {
      stale_p += __shfl_xor(stale_p, o);
      stale_r += __shfl_xor(stale_r, o);
    }
    if (lane == 0) {
      Report* R = rep + s_team * 64 + role;
      R->ticks = now() - t0;
      R->stale_part = stale_p;
      R->stale_result = stale_r;
      R->timeouts = tmo;
      R->rounds_done = (unsigned long long)(r - 1);
    }
  }
  if (acc == -1.2345f) sink[0] = acc;
}

template <int ST, int LD>
int run_pingpong(const char* name, unsigned* d_flags, unsigned long long* d_out, unsigned* d_xcc, int a, int b) {
  const int rounds = 2000;
  CK(hipMemset(d_flags, 0, 1024));
  hipLaunchKernelGGL((k_pingpong<ST, LD>), dim3(16), dim3(64), 0, 0, d_flags, a, b, rounds, d_out, d_xcc);
  CK(hipDeviceSynchronize());
  unsigned long long h[2];
  unsigned x[2];
  CK(hipMemcpy(h, d_out, sizeof h, hipMemcpyDeviceToHost));
  CK(hipMemcpy(x, d_xcc, sizeof x, hipMemcpyDeviceToHost));
  if (h[1])
    printf("  %-34s blocks %d (XCD %u) <-> %d (XCD %u): TIMEOUT (value never seen: stale)\n", name, a, x[0], b, x[1]);
  else
    printf("  %-34s blocks %d (XCD %u) <-> %d (XCD %u): %.0f ns per one-way hop\n", name, a, x[0], b, x[1],
           (double)h[0] * 10.0 / (2.0 * rounds));
  return 0;
}

template <int ST, int LD, bool AG>
int run_team(const char* name, int NB, int filler, int streams, Team* d_teams, Report* d_rep, float* d_junk, float* d_sink) {
  const int rounds = 400;
  std::vector<hipStream_t> st(streams);
  for (auto& s : st) CK(hipStreamCreate(&s));
  const size_t team_bytes = sizeof(Team) * 8, rep_bytes = sizeof(Report) * 8 * 64;
  CK(hipMemset(d_teams, 0, team_bytes * streams));
  CK(hipMemset(d_rep, 0, rep_bytes * streams));
  CK(hipDeviceSynchronize());
  for (int s = 0; s < streams; s++)
    hipLaunchKernelGGL((k_team<ST, LD, AG>), dim3(8 * NB), dim3(256), 0, st[s], d_teams + 8 * s, NB, rounds, filler, d_junk,
                       d_rep + 8 * 64 * s, d_sink);
  CK(hipDeviceSynchronize());
  std::vector<Report> h(8 * 64 * streams);
  CK(hipMemcpy(h.data(), d_rep, rep_bytes * streams, hipMemcpyDeviceToHost));
  std::vector<Team> ht(8 * streams);
  CK(hipMemcpy(ht.data(), d_teams, team_bytes * streams, hipMemcpyDeviceToHost));
  unsigned long long sp = 0, sr = 0, tmo = 0, tmax = 0, members_min = ~0ull, members_max = 0, rd_min = ~0ull;
  for (int s = 0; s < streams; s++)
    for (int x = 0; x < 8; x++) {
      members_min = std::min<unsigned long long>(members_min, ht[8 * s + x].n_members);
      members_max = std::max<unsigned long long>(members_max, ht[8 * s + x].n_members);
      for (int b = 0; b < NB; b++) {
        const Report& R = h[(size_t)(8 * s + x) * 64 + b];
        sp += R.stale_part;
        sr += R.stale_result;
        tmo += R.timeouts;
        tmax = std::max(tmax, R.ticks);
        rd_min = std::min(rd_min, R.rounds_done);
      }
    }
  printf("  %-30s NB %2d filler %5d streams %d: %.2f us per round; stale partials %llu, stale results %llu, timeouts %llu, "
         "rounds done (min) %llu, members per XCD %llu..%llu\n",
         name, NB, filler, streams, (double)tmax * 0.01 / (rounds - 1), sp, sr, tmo, rd_min, members_min, members_max);
  for (auto& s : st) CK(hipStreamDestroy(s));
  return 0;
}

int main() {
  unsigned* d_xcc;
  CK(hipMalloc(&d_xcc, 4 * 4096 * 4));
  // ---- part 1
  printf("part 1: placement census (block b expected on XCD b %% 8)\n");
  for (int R : {4, 32, 128}) {
    const int n = 8 * R;
    hipLaunchKernelGGL(k_census, dim3(n), dim3(256), 0, 0, d_xcc, 2000);
    CK(hipDeviceSynchronize());
    std::vector<unsigned> h(n);
    CK(hipMemcpy(h.data(), d_xcc, 4 * n, hipMemcpyDeviceToHost));
    int bad = 0;
    int per[8] = {0};
    for (int b = 0; b < n; b++) {
      bad += h[b] != (unsigned)(b % 8);
      if (h[b] < 8) per[h[b]]++;
    }
    printf("  grid %4d alone: %d blocks off b %% 8; per XCD %d %d %d %d %d %d %d %d\n", n, bad, per[0], per[1], per[2], per[3],
           per[4], per[5], per[6], per[7]);
  }
  {
    hipStream_t st[4];
    for (auto& s : st) CK(hipStreamCreate(&s));
    const int n = 8 * 32;
    for (int rep = 0; rep < 3; rep++) {
      for (int s = 0; s < 4; s++) hipLaunchKernelGGL(k_census, dim3(n), dim3(256), 0, st[s], d_xcc + 4096 * s, 20000);
      CK(hipDeviceSynchronize());
      std::vector<unsigned> h(4 * 4096);
      CK(hipMemcpy(h.data(), d_xcc, 4 * 4 * 4096, hipMemcpyDeviceToHost));
      for (int s = 0; s < 4; s++) {
        int bad = 0, per[8] = {0};
        for (int b = 0; b < n; b++) {
          bad += h[4096 * s + b] != (unsigned)(b % 8);
          if (h[4096 * s + b] < 8) per[h[4096 * s + b]]++;
        }
        printf("  4 concurrent grids of %d, rep %d stream %d: %d off; per XCD %d %d %d %d %d %d %d %d\n", n, rep, s, bad, per[0],
               per[1], per[2], per[3], per[4], per[5], per[6], per[7]);
      }
    }
    for (auto& s : st) CK(hipStreamDestroy(s));
  }
  // ---- part 2
  unsigned* d_flags;
  unsigned long long* d_out;
  CK(hipMalloc(&d_flags, 1024));
  CK(hipMalloc(&d_out, 64));
  printf("part 2: ping-pong, one lane per block\n");
  for (int other : {8, 1}) {  // same XCD, then the next XCD
    run_pingpong<ST_SC1, LD_SC1>("sc1 store + sc1 load", d_flags, d_out, d_xcc, 0, other);
    run_pingpong<ST_SC1, LD_SC0SC1>("sc1 store + sc0 sc1 load", d_flags, d_out, d_xcc, 0, other);
    run_pingpong<ST_PLAIN, LD_SC1>("plain store + sc1 load", d_flags, d_out, d_xcc, 0, other);
    run_pingpong<ST_PLAIN, LD_NT>("plain store + nt load", d_flags, d_out, d_xcc, 0, other);
    run_pingpong<ST_PLAIN, LD_ATOMIC_WG>("plain store + wg atomic-or poll", d_flags, d_out, d_xcc, 0, other);
    run_pingpong<ST_PLAIN, LD_PLAIN>("plain store + plain load", d_flags, d_out, d_xcc, 0, other);
  }
  // ---- part 3
  Team* d_teams;
  Report* d_rep;
  float *d_junk, *d_sink;
  CK(hipMalloc(&d_teams, sizeof(Team) * 8 * 4));
  CK(hipMalloc(&d_rep, sizeof(Report) * 8 * 64 * 4));
  CK(hipMalloc(&d_junk, sizeof(float) * 65536 * 64));
  CK(hipMemset(d_junk, 0, sizeof(float) * 65536 * 64));
  CK(hipMalloc(&d_sink, 64));
  printf("part 3: reduce + broadcast among the NB blocks of every XCD (8 teams per launch)\n");
  for (int NB : {16, 32}) {
    for (int filler : {0, 4096}) {
      for (int streams : {1, 4}) {
        if (8 * NB * streams > 1024) continue;  // 4 blocks of 256 threads per CU at most: everything must be resident
        run_team<ST_SC1, LD_SC1, true>("sc1 / sc1 / agent atomic", NB, filler, streams, d_teams, d_rep, d_junk, d_sink);
        run_team<ST_PLAIN, LD_SC1, true>("plain / sc1 / agent atomic", NB, filler, streams, d_teams, d_rep, d_junk, d_sink);
        run_team<ST_PLAIN, LD_SC1, false>("plain / sc1 / wg atomic", NB, filler, streams, d_teams, d_rep, d_junk, d_sink);
        run_team<ST_PLAIN, LD_NT, false>("plain / nt / wg atomic", NB, filler, streams, d_teams, d_rep, d_junk, d_sink);
      }
    }
  }
  return 0;
}
