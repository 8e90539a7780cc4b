// cvo_k_coeff.h -- k_coeff: compute_step_size_xi + _poly_coeff per nonzero; the pair's last block runs the update.
// Part of the kernel set of cvo_kernels.h (which states the whole iteration); compiled only as part of cvo_hip.hip.
#pragma once
#include "cvo_k_assoc.h"
#include "cvo_update.h"

namespace cvo_dev {

// ------------------------------------------------------------------------------------------
// Coefficient phase: normalised twist (compute_flow host half, CvoGPU.cu:824-835) + B,C,D,E partials, one
// thread per row position, blocks of ASSOC_THREADS rows.
// ------------------------------------------------------------------------------------------
struct CoeffShared {
  BlockRedShared<4> red;
};

// one nonzero (i, j): compute_step_size_xi for target j (CvoGPU.cu:974-986) + compute_step_size_poly_coeff
// (CvoGPU.cu:1053-1078); yy is the transformed target
// t[0..3]: what the nonzero adds to B_i, C_i, D_i, E_i (the `+=` right-hand sides of CvoGPU.cu:1068-1078)
__device__ __forceinline__ void coeff_terms(const XiMats& M, const float4 x, float temp_coef, const V3 yy, float A_ij, double (&tq)[4]) {
  const V3 w{M.omega[0], M.omega[1], M.omega[2]};
  const V3 c = cross_dev(w, yy);
  const V3 xiz{c.x + M.v[0], c.y + M.v[1], c.z + M.v[2]};
  V3 t = matvec_dev(M.m2, yy);
  const V3 xi2z{t.x + M.ohv.x, t.y + M.ohv.y, t.z + M.ohv.z};
  t = matvec_dev(M.m3, yy);
  const V3 xi3z{t.x + M.m2v.x, t.y + M.m2v.y, t.z + M.m2v.z};
  t = matvec_dev(M.m4, yy);
  const V3 xi4z{t.x + M.m3v.x, t.y + M.m3v.y, t.z + M.m3v.z};
  const float normxiz2 = dot3_dev(xiz.x, xiz.y, xiz.z, xiz.x, xiz.y, xiz.z);
  const float xiz_dot_xi2z = -dot3_dev(xiz.x, xiz.y, xiz.z, xi2z.x, xi2z.y, xi2z.z);
  const float epsil_const = __builtin_fmaf(2.0f, dot3_dev(xiz.x, xiz.y, xiz.z, xi3z.x, xi3z.y, xi3z.z),
                                           dot3_dev(xi2z.x, xi2z.y, xi2z.z, xi2z.x, xi2z.y, xi2z.z));
  const float dfx = x.x - yy.x, dfy = x.y - yy.y, dfz = x.z - yy.z;
  const float beta_ij = (float)(-2.0 * temp_coef * (double)dot3_dev(xiz.x, xiz.y, xiz.z, dfx, dfy, dfz));
  const float gamma_ij =
      (-temp_coef) * (normxiz2 + dot3_dev(2.0f * xi2z.x, 2.0f * xi2z.y, 2.0f * xi2z.z, dfx, dfy, dfz));
  const float delta_ij =
      (float)(2.0 * temp_coef * (double)(xiz_dot_xi2z + dot3_dev(-xi3z.x, -xi3z.y, -xi3z.z, dfx, dfy, dfz)));
  const float epsil_ij =
      (-temp_coef) * (epsil_const + dot3_dev(2.0f * xi4z.x, 2.0f * xi4z.y, 2.0f * xi4z.z, dfx, dfy, dfz));
  tq[0] = (double)(A_ij * beta_ij);
  tq[1] = (double)A_ij * ((double)gamma_ij + (double)(beta_ij * beta_ij) / 2.0);
  // beta^3 / 6.0 (CvoGPU.cu:1072): the IEEE division with its constant half folded (rcp_refined / div_by, cvo_device.h)
  tq[2] = (double)A_ij * ((double)__builtin_fmaf(beta_ij, gamma_ij, delta_ij) +
                         div_by((double)(beta_ij * beta_ij * beta_ij), 6.0, rcp_refined(6.0)));
  tq[3] = (double)A_ij * ((double)__builtin_fmaf(beta_ij, delta_ij, epsil_ij) +
                         1 / 2.0 * beta_ij * beta_ij * gamma_ij + 1 / 2.0 * gamma_ij * gamma_ij +
                         1 / 24.0 * beta_ij * beta_ij * beta_ij * beta_ij);
}
__device__ __forceinline__ void coeff_entry(const XiMats& M, const float4 x, float temp_coef, const V3 yy, float A_ij,
                                            double& Bi, double& Ci, double& Di, double& Ei) {
  double t[4];
  coeff_terms(M, x, temp_coef, yy, A_ij, t);
  Bi += t[0];
  Ci += t[1];
  Di += t[2];
  Ei += t[3];
}

// Rows of this block, in two steps so that the first loads of the row loop (count -> first ELL entry -> its target:
// three dependent round trips) are in flight while the twist is reduced.
struct CoeffRowHead {
  bool dense;  // the row is beyond its cached list: k_coeff_dense has left its sums (PairDesc::rowcoef)
  unsigned nnz;
  float4 x;
  EllEntry e_n;  // the row's first entry of this block's slice
};
// COH: the block partial is read by another block of the same launch.
template <bool COH>
__device__ __forceinline__ void coeff_rows(const DevParams& P, const PairDesc* __restrict__ D, const float ell,
                                           const float coef_ell, CoeffShared& S, const XiMats& Mu, const CoeffRowHead& h, const int bx,
                                           const int q, const int nsplit, const Pose& pose, const unsigned tag) {
  const int N = D->N;
  const int i = bx * ASSOC_THREADS + threadIdx.x;
  double Bi = 0, Ci = 0, Di = 0, Ei = 0;
  // this block's share of the row: slots q, q + nsplit, ...  (small clouds whose rows sit on K_max would
  // otherwise leave the chip to a handful of waves walking hundreds of entries each).  Software pipeline: the
  // next entry's index / value / target are in flight while the current one is evaluated.
  const unsigned nnz = h.nnz;
  if (h.dense) {
    // a row k_assoc_dense and k_coeff_dense evaluated (a wave per row): its sums, slice by slice in this kernel's own order
    if (i < N) {
      const double* rc = D->rowcoef + ((size_t)i * nsplit + q) * 4;
      Bi = rc[0];
      Ci = rc[1];
      Di = rc[2];
      Ei = rc[3];
    }
  } else if ((unsigned)q < nnz) {
    const float4 x = h.x;
    // 1 / (2.0 * ell * ell), CvoGPU.cu:1060: the same for every row (PairState::temp_coef, evaluated by the update when
    // ell changes) unless the range factor is on (CvoGPU.cu:1035-1037)
    float temp_coef = coef_ell;
    if (P.use_range_ell) {
      const float d2_sqrt = sqrtf(dot3_dev(x.x, x.y, x.z, x.x, x.y, x.z));
      temp_coef = coef_of_ell(compute_range_ell(ell, d2_sqrt));
    }
#ifdef CVO_ELL8
    // 8-byte entries: the initial target is gathered again by its sorted position and transformed with the iteration's
    // pose - two loads deep: entry s + 2 nsplit and target s + nsplit are in flight while entry s is evaluated
    const CVO_GLOBAL f32x4* ysrc = (const CVO_GLOBAL f32x4*)D->ys4;
    const CVO_GLOBAL f32x2* ep = (const CVO_GLOBAL f32x2*)D->ell + i;
    auto ld_e = [](const CVO_GLOBAL f32x2* p) { const f32x2 v = *p; return EllEntry{v.x, __float_as_int(v.y)}; };
    EllEntry e_n = h.e_n;
    float4 y_n = ldg_xyz(ysrc + e_n.p);
    EllEntry e_nn = e_n;
    if ((unsigned)q + nsplit < nnz) e_nn = ld_e(ep + (size_t)(q + nsplit) * N);
    for (unsigned s = (unsigned)q; s < nnz; s += (unsigned)nsplit) {
      const EllEntry e = e_n;
      const float4 y0 = y_n;
      e_n = e_nn;
      if (s + nsplit < nnz) y_n = ldg_xyz(ysrc + e_n.p);
      if (s + 2 * nsplit < nnz) e_nn = ld_e(ep + (size_t)(s + 2 * nsplit) * N);
      const V3 yy = transform_point(pose.Ri, pose.Ti, y0.x, y0.y, y0.z);
      coeff_entry(Mu, x, temp_coef, yy, e.a, Bi, Ci, Di, Ei);
    }
#else
    EllEntry e_n = h.e_n;
    for (unsigned s = (unsigned)q; s < nnz; s += (unsigned)nsplit) {
      const EllEntry e = e_n;
      if (s + nsplit < nnz) e_n = D->ell[(size_t)(s + nsplit) * N + i];  // next entry in flight
      // (the transformed target k_assoc evaluated the pair with: transform_point of the same operands, stored)
      coeff_entry(Mu, x, temp_coef, V3{e.yx, e.yy, e.yz}, e.a, Bi, Ci, Di, Ei);
    }
#endif
  }
  const double red[4] = {Bi, Ci, Di, Ei};
  const double tot = block_reduce_lds<4>(S.red, red);
  // (data-tagged granules, cvo_wave.h: read by the pair's updating block without anybody waiting for a store to be acknowledged)
  if (threadIdx.x < 32 && (threadIdx.x & 7) == 0 && !(P.debug_drop_partial == 2 && bx == 1))
    st_tagged(D->coef_part + ((size_t)bx * nsplit + q) * COEF_GRANULES + 2 * (threadIdx.x >> 3), tot, tag);
}

// ------------------------------------------------------------------------------------------
// k_coeff: coefficient phase + (align loop) the update.  The block of a pair that finishes last runs update_body:
// one launch less on the critical path of every iteration, and no block ever waits for another one.  Partials
// cross blocks inside the launch, hence the coherent stores / loads (st_x / ld_x).
// flags: bit 0 = lean graph, bit 5 = ... with k_assoc_dense in every iteration, the rest see update_body.
// ------------------------------------------------------------------------------------------
template <bool INSTR>
__global__ __launch_bounds__(ASSOC_THREADS, CVO_COEFF_WAVES) void k_coeff(const PairDesc* __restrict__ descs,
                                                         const DevParams* __restrict__ Pp, PairState* states,
                                                         const char* __restrict__ arena, int flags, int nblk_split_pairs,
                                                         unsigned stride256, int Npad) {
  const unsigned long long tt0 = INSTR ? __builtin_readcyclecounter() : 0ull;
  // grid: per pair nblk row blocks x launch_split slices of the ELL slots; a pair uses csplit <= launch_split of them
  const int nblk = nblk_split_pairs & 0x3fff, launch_split = (nblk_split_pairs >> 14) & 0x3f,
            n_pairs = (int)((unsigned)nblk_split_pairs >> 20);
  // (+ 1: the pair's speculative block, update_speculate)
  PairBlock pb;
  if (!pair_block(nblk * launch_split + 1, n_pairs, pb)) return;
  const PairDesc* __restrict__ D = descs + pb.pair;
  __shared__ union {
    CoeffShared c;
    UpdateShared u;
  } S;
#ifndef CVO_SPEC_BLOCK_LAST
  // (the pair's FIRST block: workgroups are dispatched in index order, and this run has the row blocks' time and no more -
  // as the last block of its pair it started late on a full chip and was adopted less often)
  pb.bx -= 1;
  if (pb.bx < 0) {
#else
  if (pb.bx == nblk * launch_split) {
#endif
    // The speculative block: everything of the update that follows the step, on the predicted step, while the row blocks
    // work.  Not in the instrumented kernels, timing replays, traced calls (the record needs B..E) or single evaluations;
    // a waiting pair (lean graph) has nothing to advance.
    if (INSTR || threadIdx.x >= 64 || (flags & (8 | 16))) return;
    const PairState* __restrict__ sh = states + pb.pair;
    const int status_h = sh->status, rebuild_h = sh->rebuild, ovf_h = sh->n_ovf;
    const DevParams Ph = *Pp;
    if (status_h != 0 || Ph.mode != 0 || Ph.trace_capacity != 0) return;
    if ((flags & 1) && (rebuild_h || (ovf_h > 0 && !(flags & 32)))) return;
    update_speculate(D, states + pb.pair, Ph, flags | 4, S.u, D->call_serial);
    return;
  }
  const int cq = pb.bx % launch_split;
  pb.bx /= launch_split;
  // head of the row loop, from kernel-argument addresses (row_off_*): count, coordinates and the first ELL entry
  // of this block's slice - requested before the count is known, used only if it exists
  CoeffRowHead head;
  {
    const char* wb = arena + (size_t)pb.pair * ((size_t)stride256 << 8);
    const int pos = pb.bx * ASSOC_THREADS + threadIdx.x;  // < Npad; values of rows >= N are never used
    head.nnz = nnz_count(reinterpret_cast<const unsigned*>(wb + row_off_nnz(Npad))[pos]);
    head.x = reinterpret_cast<const float4*>(wb + row_off_xp4(Npad))[pos];
    head.dense = false;
    head.e_n = make_ell(0.f, 0.f, 0.f, 0.f, 0);
    if (cq == 0) head.e_n = reinterpret_cast<const EllEntry*>(wb + row_off_ell(Npad))[pos];
  }
  const int csplit = D->csplit;
  PairState* const st = states + pb.pair;  // == D->st, without the dependent pointer load
  // The state as this launch found it, through a read-only view so that the loads are scalar (only the block that
  // finishes last writes the state, after every block has read it); one burst together with what the row loop
  // needs first, see k_assoc.
  const PairState* __restrict__ st_in = states + pb.pair;
  const int status_v = st_in->status, rebuild_v = st_in->rebuild, ovf = st_in->n_ovf;
  const int row_max_v = st_in->row_max;
  const DevParams P = *Pp;
  // the twist and its matrices (twist_finalize): wave-uniform scalar loads
  XiMats Mu;
  {
    float* mu = reinterpret_cast<float*>(&Mu);
#pragma unroll
    for (int q = 0; q < (int)(sizeof(XiMats) / sizeof(float)); q++) mu[q] = st_in->xi[q];
  }
  {
    // (everything the kernel will branch on or start its row loop with - the slice count, the parameters and the twist
    // matrices included - requested before the first wait: each dependent round of scalar loads is ~0.3-0.5 us here)
    const int n = D->N, nb = D->nblk_assoc, ep = st_in->epoch;
    const unsigned long long* a0 = D->flow_part;
    const unsigned* a1 = D->nnz_row;
    const float4* a2 = D->xp4;
    const EllEntry* a3 = D->ell;
    const int a4 = D->M;
    const float e = st_in->ell, tc = st_in->temp_coef;
    const int k_line = st_in->K;  // (rides in the 16-byte load of status / rebuild / n_ovf: pinned so that none of its
                                  // registers is dead and reused inside the burst, see k_assoc)
    asm volatile("" ::"s"(n), "s"(nb), "s"(ep), "s"(a0), "s"(a1), "s"(a2), "s"(a3), "s"(a4), "s"(e), "s"(tc), "s"(csplit),
                 "s"(row_max_v), "s"(status_v), "s"(rebuild_v), "s"(ovf), "s"(k_line), "s"(P.mode), "s"(P.sp_thres), "s"(P.use_range_ell),
                 "s"(Mu.omega[0]), "s"(Mu.m2.m[0][0]), "s"(Mu.m4.m[2][2]), "s"(Mu.v[2]));
    // (nothing computed from these values - the slice count below is the first - may be scheduled into the middle of
    // the burst, where it would need a wait of its own: one more round trip)
    __builtin_amdgcn_sched_barrier(0);
  }
  // (rows beyond their cached lists - hundreds of nonzeros: clustered clouds, the K cap - are not walked here: k_coeff_dense,
  // a wave per row, has left their sums in PairDesc::rowcoef; the split is the pair's own, whatever company it is solved in)
  if (cq >= csplit) return;
  const bool replay = (flags & 8) != 0;  // cvo_debug_time_kernels: same work, nothing written back
  if (!replay && status_v != 0) return;
  if (flags & 1) {
    if (rebuild_v || (ovf > 0 && !(flags & 32))) {  // waiting, see k_assoc; tell the host which graph this pair needs
      if (pb.bx == 0 && cq == 0 && threadIdx.x == 0) {
        st->n_stalls++;
        if (ovf > 0 && !(flags & 32)) {
          st->want_full = 4;
          *D->want_out = 4;
          *D->want_host = 4;
        }
      }
      return;
    }
  }
  if (P.mode != 0) return;
  pair_clock_begin(INSTR && P.kernel_clock && !replay && pb.bx == 0 && cq == 0, st, 1);
  const int epoch = st_in->epoch;  // launches of this kernel the pair has completed (bumped by the updating block)
  __shared__ int s_last;
  const int N_ = D->N, pos_ = pb.bx * ASSOC_THREADS + threadIdx.x;
  if (pos_ >= N_) head.nnz = 0;
  // the row's class as k_list and k_assoc see it: more candidates than row_max (bit 6: 32-entry lists, M >= 65536)
  head.dense = pos_ < N_ && ovf > 0 && __float_as_int(head.x.w) > min(row_max_v, (flags & 64) ? ASSOC_CAP32 : ASSOC_CAP16);
  if (cq > 0 && (unsigned)cq < head.nnz) head.e_n = D->ell[(size_t)cq * N_ + pos_];  // (small clouds only: later slices)
  float twist[6];
  for (int c = 0; c < 3; c++) {
    twist[c] = Mu.omega[c];
    twist[3 + c] = Mu.v[c];
  }
  const unsigned long long tt1 = INSTR ? __builtin_readcyclecounter() : 0ull;
#ifdef CVO_ELL8
  const Pose pose = load_pose(st_in);
#else
  const Pose pose{};
#endif
  // tag of this launch's partials: the call's serial and the pair's launch generation (a timing replay re-uses the tag of the
  // launch it replays: same values, nothing is written back)
  const unsigned tag = partial_tag(D->call_serial, 0x80000000u | (unsigned)epoch);
  coeff_rows<true>(P, D, st_in->ell, st_in->temp_coef, S.c, Mu, head, pb.bx, cq, csplit, pose, tag);
  const unsigned long long tt2 = INSTR ? __builtin_readcyclecounter() : 0ull;
  if (threadIdx.x >= 64) return;  // the counter and (in one block of the pair) the update are the first wave's, see k_assoc
  // the scalar state, for whichever block turns out to be the last one: in flight while the counter round trip runs
  unsigned hot_regs[2] = {0u, 0u};
  if (threadIdx.x < 64) {
    hot_regs[0] = reinterpret_cast<const unsigned*>(st)[threadIdx.x];
    if (threadIdx.x + 64 < HOT_DWORDS) hot_regs[1] = reinterpret_cast<const unsigned*>(st)[threadIdx.x + 64];
  }
  const unsigned long long clk0 = pair_clock_peek(INSTR && P.kernel_clock && !replay, st, 1);
  UpdDesc upd = load_upd_desc(D);
  upd.nblk_coeff = nblk * csplit;
  const int n_flow_upd = D->nblk_assoc;  // (requested now, not after the counter's round trip on the pair's serial tail)
  asm volatile("" ::"s"(n_flow_upd));
  __syncthreads();  // (no wait for the partial's store: its granules carry the launch's tag, see flow_gate)
  if (threadIdx.x == 0) {
    // the counter advances by nblk * COEFF_SPLIT_MAX per iteration whatever the split of the iteration is (splits are
    // powers of two): each of the nblk * csplit blocks that store a partial adds its share
    const unsigned share = (unsigned)COEFF_SPLIT_MAX >> __builtin_ctz((unsigned)csplit), per_it = (unsigned)(nblk * COEFF_SPLIT_MAX);
    const unsigned done = (unsigned)__hip_atomic_fetch_add(D->done, (int)share, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + share;
    s_last = replay ? (done % per_it == 0u) : (done == (unsigned)(epoch + 1) * per_it);
  }
  __syncthreads();
  const unsigned long long tt3 = INSTR ? __builtin_readcyclecounter() : 0ull;
  if (INSTR && P.phase_ticks && threadIdx.x == 0) {
    g_phase_ticks[1][blockIdx.x & 4095][0] = tt0;
    g_phase_ticks[1][blockIdx.x & 4095][1] = tt1;
    g_phase_ticks[1][blockIdx.x & 4095][2] = tt2;
    g_phase_ticks[1][blockIdx.x & 4095][3] = tt3;
  }
  if (!s_last || (flags & 16)) return;  // (bit 4: cost breakdown of cvo_debug_time_kernels, coefficient phase only)
  update_body<false, true>(upd, P, flags | 4, n_flow_upd, S.u, twist, hot_regs, clk0, tag, !INSTR);
  if (INSTR && P.phase_ticks && threadIdx.x == 0) {
    g_phase_ticks[1][4096 + pb.pair][0] = tt0;
    g_phase_ticks[1][4096 + pb.pair][1] = tt3;
    g_phase_ticks[1][4096 + pb.pair][2] = __builtin_readcyclecounter();
  }
}

}  // namespace cvo_dev
