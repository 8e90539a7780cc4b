// cvo_k_debug.h -- k_verify (CVO_VERIFY_LISTS), k_scalar_math (device scalar maths on caller inputs), k_hold.
// Part of the kernel set of cvo_kernels.h (which states the whole iteration); compiled only as part of cvo_hip.hip.
#pragma once
#include "cvo_pair_math.h"
#include "cvo_update.h"

namespace cvo_dev {

// ------------------------------------------------------------------------------------------
// k_verify (CVO_VERIFY_LISTS=1): the self-check of the candidate-list reuse.  After the association of an iteration
// (k_assoc over the cached lists [+ k_assoc_dense]) one wave per row re-derives the row with the reference's literal
// ordered scan over ALL targets (CvoGPU.cu:522-590) at the pose / ell / K of that iteration and compares it with the
// row the lists produced: nonzero count, every column, every value bit for bit.  A list that had lost a pair - a skin
// too small for the motion since the build, a cull that was not conservative - shows up as a missing or shifted entry.
// The first mismatch of a pair is latched in its state (sticky) and turns the call's return code into CVO_E_VERIFY.
// Independent of the oracle and of the clouds' size: the tests run it at 10k x 10k over the fast-moving first iterations.
// ------------------------------------------------------------------------------------------
template <int FEAT>
__global__ __launch_bounds__(256) void k_verify(const PairDesc* __restrict__ descs, const DevParams* __restrict__ Pp,
                                                const int* __restrict__ status, int lean) {
  if (status[blockIdx.y] != 0) return;
  const PairDesc* __restrict__ D = descs + blockIdx.y;
  PairState* st = D->st;
  if ((lean & 1) && (st->rebuild || (st->n_ovf > 0 && !(lean & 4)))) return;  // the pair did not advance in this slot (see k_assoc)
  const DevParams P = *Pp;
  const int N = D->N, M = D->M, K = st->K;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const Pose pose = load_pose(st);
  const FeatDen F = make_feat_den(P);
  unsigned checked = 0;
  for (int pos = blockIdx.x * 4 + wave; pos < N; pos += gridDim.x * 4) {
    const int i = D->ip[pos];
    const float4 x = D->xp4[pos];
    const RowData r = make_row(P, x, st->ell);
    unsigned nnz = 0;
    int err = 0;
    const unsigned nnz_word = D->nnz_row[pos];  // (flagged: the wave-per-row kernels evaluated the row, its entries may lie row-major)
    const int off = (nnz_word & NNZ_DENSE_FLAG) ? D->dense_off[pos] : -1;
    for (int j0 = 0; j0 < M && nnz < (unsigned)K; j0 += 64) {
      const int j = j0 + lane;
      float a = 0.f;
      float4 yt;
      bool ok = false;
      if (j < M) ok = eval_pair<FEAT>(P, D, F, pose, i, r, FEAT != FEAT_GEO ? D->yinv[j] : 0, D->y4[j], a, yt) && (a > P.sp_thres);
      const unsigned long long m = __ballot(ok);
      const unsigned rank = nnz + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
      const bool keep = ok && rank < (unsigned)K;
      if (keep) {
        const EllEntry e = D->ell[ell_index(N, (int)rank, pos, off)];
        if (D->ell_j[(size_t)rank * N + pos] != j)
          err = 2;
#ifdef CVO_ELL8
        else if (__float_as_uint(e.a) != __float_as_uint(a) || D->yorder[e.p] != j)
          err = 3;
#else
        else if (__float_as_uint(e.a) != __float_as_uint(a) || __float_as_uint(e.yx) != __float_as_uint(yt.x) ||
                 __float_as_uint(e.yy) != __float_as_uint(yt.y) || __float_as_uint(e.yz) != __float_as_uint(yt.z))
          err = 3;
#endif
      }
      nnz += (unsigned)__builtin_popcountll(__ballot(keep));
    }
    if (nnz_count(nnz_word) != nnz) err = 1;  // (also catches entries the list path has and the scan does not)
    if (__ballot(err != 0) != 0ull) {
      int e = err;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) e = max(e, __shfl_xor(e, o));
      if (lane == 0 && atomicCAS(&st->verify_err, 0, 1) == 0) {
        st->verify_k = st->k;
        st->verify_pos = pos;
        st->verify_what = (nnz_count(nnz_word) != nnz) ? 1 : e;
      }
    }
    checked++;
  }
  if (lane == 0 && checked) atomicAdd(&st->verify_rows, (unsigned long long)checked);
}

// ------------------------------------------------------------------------------------------
// k_scalar_math (cvo_debug_scalar_math): the device's scalar restatements of the reference's host-side maths, run
// on caller-supplied inputs so that the tests can pin THE DEVICE CODE ITSELF against numpy / scipy (the oracle
// carries the same text for some of them, so "GPU == oracle" alone only shows that two compilers agree).
// One wave per item; item q reads in[16 q ..] and writes out[16 q ..].
//   op 0  cubic_roots            in: p0..p3                      out: re[3], im[3]
//   op 1  cubic_roots_wave       (the three-lane search used by the update)  same layout
//   op 2  select_step<false>     in: B, C, D, E, min_step, max_step          out: step
//   op 3  select_step<true>      same; out[1] = 1 / 2 when the certified Newton shortcut answered, 0 for the full solve
//   op 13 select_step<true, false>  the full solve alone (the shortcut's answers must equal it bit for bit)
//   op 4  exp_sek3               in: xi[6], dt                   out: 3x4 row-major
//   op 5  se3_log_norm           in: R[9] row-major, t[3]        out: norm
//   op 6  update_tf              in: R[9], T[3]                  out: Rinv[9], Tinv[3]
//   op 7  indicator windows      ONE item: in = {window, threshold, x_0 .. x_{n-1}}, out[k] = decision of sample k
//                                (indicator_update on a scratch PairState, exactly as the update calls it)
// The hoisted arithmetic of the row loops against the compiler's / the device library's own forms (eight operands per
// item, lane l takes operand l; out[2 l] = the plain form, out[2 l + 1] = the hoisted form - the tests compare the BITS):
//   op 8  n / d  vs  div_by(n, d, rcp_refined(d))            in: {n_l, d_l} pairs (in[2 l], in[2 l + 1])
//   op 9  x / 6.0  vs  div_by(x, 6.0, rcp_refined(6.0))      in: x_l (in[l])
//   op 10 exp(x)  vs  exp_ocml<false>(x)                     in: x_l
//   op 11 exp(x)  vs  exp_ocml<true>(x)   (x <= 0)           in: x_l
//   op 12 (float) n / d  vs  fdiv_hoisted(n, fdiv_prepare(d)) in: {n_l, d_l} pairs; out[2 l + 1] = NaN-boxed -1 (as a
//         double: -1.0) where fdiv_operands_safe refuses the operand (the kernel then divides the plain way)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_scalar_math(int op, int n, const double* __restrict__ in, double* __restrict__ out,
                                                    PairState* scratch) {
  const int lane = threadIdx.x;
  if (op == 7) {
    if (blockIdx.x != 0 || lane != 0) return;
    const int window = (int)in[0];
    const float thr = (float)in[1];
    for (int k = 0; k < n; k++) {
      const float e_front = scratch->eq[scratch->e_head], s_front = scratch->sq[scratch->s_head];
      out[k] = indicator_update(scratch, scratch->sq, scratch->eq, (float)in[2 + k], window, thr, e_front, s_front) ? 1.0 : 0.0;
    }
    return;
  }
  const double* a = in + 16 * (size_t)blockIdx.x;
  double* o = out + 16 * (size_t)blockIdx.x;
  if (op == 12) {
    if (lane >= 8) return;
    const float nn = (float)a[2 * lane], dd = (float)a[2 * lane + 1];
    const float six[6] = {nn, 0.f, 0.f, 0.f, 0.f, 0.f};
    o[2 * lane] = (double)(nn / dd);
    o[2 * lane + 1] = fdiv_operands_safe(six) ? (double)fdiv_hoisted(nn, fdiv_prepare(dd)) : -1.0;
    return;
  }
  if (op >= 8 && op <= 11) {
    if (lane >= 8) return;
    double plain, hoisted;
    if (op == 8) {
      const double nn = a[2 * lane], dd = a[2 * lane + 1];
      plain = nn / dd;
      hoisted = div_by(nn, dd, rcp_refined(dd));
    } else if (op == 9) {
      const double x = a[lane];
      plain = x / 6.0;
      hoisted = div_by(x, 6.0, rcp_refined(6.0));
    } else {
      const double x = a[lane];
      const ExpConsts ek = make_exp_consts();
      plain = exp(x);
      hoisted = op == 10 ? exp_ocml<false>(x, ek) : exp_ocml<true>(x, ek);
    }
    o[2 * lane] = plain;
    o[2 * lane + 1] = hoisted;
    return;
  }
  if (op == 0 || op == 1) {
    const double coef[4] = {a[0], a[1], a[2], a[3]};
    double re[3], im[3];
    if (op == 0)
      cubic_roots(coef, re, im);
    else
      cubic_roots_wave(coef, re, im);
    if (lane == 0)
      for (int q = 0; q < 3; q++) {
        o[q] = re[q];
        o[3 + q] = im[q];
      }
  } else if (op == 2 || op == 3) {
    int path = 0;
    const float st = op == 2 ? select_step<false>(a[0], a[1], a[2], a[3], (float)a[4], (float)a[5], &path)
                             : select_step<true>(a[0], a[1], a[2], a[3], (float)a[4], (float)a[5], &path);
    if (lane == 0) {
      o[0] = (double)st;
      o[1] = (double)path;  // 1 / 2: the certified Newton shortcut answered (root / max_step), 0: the full solve
    }
  } else if (op == 13) {  // the full solve alone, to compare bits with op 3
    const float st = select_step<true, false>(a[0], a[1], a[2], a[3], (float)a[4], (float)a[5]);
    if (lane == 0) o[0] = (double)st;
  } else if (op == 4) {
    float xi[6], dt = (float)a[6], res[12];
    for (int q = 0; q < 6; q++) xi[q] = (float)a[q];
    exp_sek3(xi, dt, res);
    if (lane == 0)
      for (int q = 0; q < 12; q++) o[q] = (double)res[q];
  } else if (op == 5) {
    double R[9], t[3];
    for (int q = 0; q < 9; q++) R[q] = a[q];
    for (int q = 0; q < 3; q++) t[q] = a[9 + q];
    const double v = se3_log_norm(R, t);
    if (lane == 0) o[0] = v;
  } else if (op == 6) {
    float R[9], T[3], Ri[9], Ti[3];
    for (int q = 0; q < 9; q++) R[q] = (float)a[q];
    for (int q = 0; q < 3; q++) T[q] = (float)a[9 + q];
    update_tf(R, T, Ri, Ti);
    if (lane == 0) {
      for (int q = 0; q < 9; q++) o[q] = (double)Ri[q];
      for (int q = 0; q < 3; q++) o[9 + q] = (double)Ti[q];
    }
  }
}

// Waits for `ticks` of the s_memrealtime counter (cvo_debug_kernel_clock calibrates the counter's rate with it).
__global__ void k_hold(unsigned long long ticks) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}

}  // namespace cvo_dev
