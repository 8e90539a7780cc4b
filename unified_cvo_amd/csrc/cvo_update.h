// cvo_update.h -- update_body / k_update: the reference's host-side scalar code of one iteration (step, pose, indicator, ell, K) and the candidate-list bookkeeping, on one wave.
// Part of the kernel set of cvo_kernels.h (which states the whole iteration); compiled only as part of cvo_hip.hip.
#pragma once
#include "cvo_wave.h"
#include "cvo_pair_math.h"

namespace cvo_dev {

// ------------------------------------------------------------------------------------------
// k_update: per-pair scalar bookkeeping, one wave per pair.  INIT = true is the launch before the first
// iteration (no bookkeeping, state comes from the host).
// ------------------------------------------------------------------------------------------
constexpr int HOT_DWORDS = (int)(offsetof(PairState, sq) / 4);  // the scalar part of the state (the 4 KB of indicator FIFOs stay in HBM)
constexpr int SHADOW_WORDS = 128;  // a speculative state as data-tagged granules: the staged state + UpdOut (update_speculate)
static_assert(HOT_DWORDS + 4 <= SHADOW_WORDS, "the shadow holds the scalar state and four words of UpdOut");
struct UpdateShared {
  double c[4];
  unsigned long long n[4];
  unsigned hot[HOT_DWORDS];
  unsigned ext[SHADOW_WORDS - HOT_DWORDS];  // [0] done, [1] a request word was posted, [2] its value, [3] the predicted step (bits)
};

// What the update reads from the pair descriptor, requested in one burst of scalar loads (k_coeff issues it while
// the last-block counter is on its way): every field first touched in the middle of the serial tail would be
// another cold round trip there.
struct UpdDesc {
  PairState* st;
  const unsigned long long* coef_part;
  const unsigned long long* flow_part;
  const unsigned long long* cnt_part;
  cvo_trace_t* trace;
  int* status_out;
  int* want_out;
  int* status_host;
  int* want_host;
  unsigned long long* shadow;
  int nblk_coeff, N, M, max_iter;
  float ymax;
  double sqrt_nm;
};
__device__ __forceinline__ UpdDesc load_upd_desc(const PairDesc* __restrict__ D) {
  UpdDesc u;
  u.st = D->st;
  u.coef_part = D->coef_part;
  u.flow_part = D->flow_part;
  u.cnt_part = D->cnt_part;
  u.trace = D->trace;
  u.status_out = D->status_out;
  u.want_out = D->want_out;
  u.status_host = D->status_host;
  u.want_host = D->want_host;
  u.shadow = D->shadow;
  u.nblk_coeff = D->nblk_coeff;
  u.N = D->N;
  u.M = D->M;
  u.max_iter = D->max_iter;
  u.ymax = D->ymax;
  u.sqrt_nm = D->sqrt_nm;
  asm volatile("" ::"s"(u.st), "s"(u.coef_part), "s"(u.flow_part), "s"(u.cnt_part), "s"(u.trace), "s"(u.status_out),
               "s"(u.want_out), "s"(u.nblk_coeff), "s"(u.N), "s"(u.M), "s"(u.ymax));
  return u;
}

// Executed by the first wave of the calling block (the other threads only take part in the barriers).
// flags: bit 1 = the rebuild kernels run right after this iteration, bit 2 = called from k_coeff, bit 3 = replay for
// timing (nothing is written back), bits 8.. = how many
// iterations the list has to survive without another rebuild opportunity (0 in the full graph).  n_flow_parts: association partials to
// sum (the lean graph has no k_assoc_dense, so its slots are not read).
// PairState::want_full, the graph a pair asks the host for: a LEVEL - 2 = a rebuild opportunity in every iteration,
// 1 = every lean_U2 iterations (short lean graph), 0 = every lean_U (lean graph), -1 = calm, one per chunk - and whether
// k_assoc_dense has to run (overflow rows / the dense regime).  Encoded as: level without the dense kernel; 4 = level 2
// with it; 8 + (level + 1) = a leaner level with it. 
__device__ __forceinline__ int want_level(int w) { return w == 4 ? 2 : (w >= 8 ? w - 9 : w); }
__device__ __forceinline__ int want_encode(int level, bool dense) { return !dense ? level : (level >= 2 ? 4 : 9 + level); }

// What the scalar part of an iteration leaves OUTSIDE the staged state: the status / request words the host polls.  A
// speculative run (update_speculate) only records them; the block that adopts its state posts them.
struct UpdOut {
  int done = 0, want_write = 0, want_val = 0;
  __device__ __forceinline__ void post_want(const UpdDesc& D, bool spec, int want) {
    if (spec) {
      want_write = 1;
      want_val = want;
    } else {
      *D.want_out = want;
      *D.want_host = want;
    }
  }
  __device__ __forceinline__ void post_done(const UpdDesc& D, bool spec) {
    if (spec) {
      done = 1;
    } else {
      *D.status_out = 1;
      *D.status_host = 1;
    }
  }
};

// Everything of an iteration's scalar code that follows the step: pose, distance, exits, indicator, ell / K, update_tf,
// the candidate-list bookkeeping (one lane; `st` is the state staged in LDS, s_c / s_n the reduced coefficient sums and
// counts).  spec: a speculative run on a predicted step (update_speculate) - B..E are unknown and not stored, the host
// words are recorded in `out` instead of written.
template <bool INIT>
__device__ __forceinline__ void update_advance(const UpdDesc& D, const DevParams& P, const int flags, PairState* const st,
                                               float* const sq, float* const eq, const double* s_c,
                                               const unsigned long long* s_n, const float step_w, const float* twist,
                                               const float e_front, const float s_front, const unsigned long long clk0,
                                               const bool spec, UpdOut& out) {
  const bool trio_follows = INIT || (flags & 2) != 0;
  const bool dry = (flags & 8) != 0;  // timing replay: compute everything, write nothing back
  const int horizon = flags >> 8;
  int done = 0;
  if (!INIT && st->sync_err) {  // a partial of this launch or of k_assoc's never arrived (cvo_wave.h): the pair ends here
    done = 1;
    st->iterations = st->k;
  }
  if (INIT) st->temp_coef = coef_of_ell(st->ell);
  if (twist) {  // k_coeff: every block derived the same normalised twist
    for (int c = 0; c < 3; c++) {
      st->omega[c] = twist[c];
      st->v[c] = twist[3 + c];
    }
  }
  if (!INIT) {
    if (flags & 4) st->epoch++;  // generation of k_coeff's last-block counter
    const unsigned nnz = (unsigned)s_n[0], max_nnz = (unsigned)s_n[1];
    st->nnz = nnz;
    st->max_nnz = max_nnz;
    st->ncand = st->ncand_list;  // candidates of the current lists (k_list), evaluated exactly in this iteration
    st->ncand_total += st->ncand_list;
    st->noverflow = s_n[3];
    st->K_last = st->K;  // the stride upstream wrote this iteration's A matrix with (gpu_association_to_cpu)
    if (P.mode != 0) {  // single evaluation: A_sum (SparseKernelMat.cu:62-68)
      st->asum = s_c[0];
      done = 1;
    } else {
      if (!spec) {  // (a speculative run does not know them: whoever adopts its state stores them, see update_body)
        st->B = s_c[0];
        st->C = s_c[1];
        st->D = s_c[2];
        st->E = s_c[3];
      }
      const float step = step_w;
      st->step = step;
      const int k = st->k;
      const int K_used = st->K;
      const float ell_used = st->ell;
      const float* om = st->omega;
      const float* vv = st->v;
      double dist = 0;
      auto sqnorm3d = [](const float* a) {
        const double x = a[0], y = a[1], z = a[2];
        return x * x + (y * y + z * z);
      };
      // `omega.norm() < eps && v.norm() < eps` (double sqrt of the float-derived sums).  The twist is normalised, so
      // one of the two is ~1: sqrt is monotonic and correctly rounded, x > eps^2 (1 + 1e-12) decides sqrt(x) >= eps
      // without the ~60 dependent instructions of a double square root on the serial tail (exact shortcut).
      const double n2o = sqnorm3d(om), n2v = sqnorm3d(vv);
      const double eps2_hi = (double)P.eps * (double)P.eps * (1.0 + 1e-12);
      bool vanished = false;
      if (!(n2o > eps2_hi || n2v > eps2_hi)) vanished = sqrt(n2o) < (double)P.eps && sqrt(n2v) < (double)P.eps;
      if (vanished) {  // CvoGPU.cu:1454-1458
        auto norm3f = [](const float* a) { return sqrtf(a[0] * a[0] + (a[1] * a[1] + a[2] * a[2])); };
        if ((double)norm3f(om) < 1e-8 && (double)norm3f(vv) < 1e-8) st->ret = -1;
        done = 1;
        st->iterations = k;
      } else {
        const float xi[6] = {om[0], om[1], om[2], vv[0], vv[1], vv[2]};
        float dtrans[12];
        exp_sek3(xi, step, dtrans);  // CvoGPU.cu:1462
        // (the increment stays in its twelve floats; widened where it is used: kept as doubles it held 24 registers across
        // everything up to the - rarely taken - logarithm below, and the update's registers are what caps k_coeff's occupancy)
        auto dRd = [&](int q) { return (double)dtrans[4 * (q / 3) + (q % 3)]; };
        auto dTd = [&](int i) { return (double)dtrans[4 * i + 3]; };
        // (the running pose is fetched from the staged state row by row, only now: held in registers from the top of the
        // update it was live through Exp_SEK3, where the register count of the whole kernel peaks)
        float Rc[9], Tc[3];  // (requested together, one LDS round trip; the rows below are kept apart by scheduling
        for (int q = 0; q < 9; q++) Rc[q] = st->R[q];  // barriers: one row's double temporaries at a time)
        for (int q = 0; q < 3; q++) Tc[q] = st->T[q];
#pragma unroll
        for (int i = 0; i < 3; i++) {  // CvoGPU.cu:1463-1469
          const double r0 = Rc[3 * i + 0], r1 = Rc[3 * i + 1], r2 = Rc[3 * i + 2];
          const float tn = (float)((r0 * dTd(0) + (r1 * dTd(1) + r2 * dTd(2))) + (double)Tc[i]);
          float rn[3];
          for (int j = 0; j < 3; j++) rn[j] = (float)(r0 * dRd(0 + j) + (r1 * dRd(3 + j) + r2 * dRd(6 + j)));
          st->T[i] = tn;
          for (int j = 0; j < 3; j++) st->R[3 * i + j] = rn[j];
          __builtin_amdgcn_sched_barrier(0);
        }
        // dist = || log SE3(dR, dT) || (CvoGPU.cu:1473-1476) decides one thing: dist < eps_2.  dR / dT are the float
        // Exp_SEK3 of a unit twist times `step`, so in exact arithmetic dist = step * |xi|_6 = step; the float
        // rounding of dtrans (6e-8 per entry, entries <= 1) and of the normalisation move it by < 1e-6 + 1e-4 step.
        // When step clears eps_2 by that margin the comparison is decided and the ~300 dependent double-precision
        // instructions of the log (quaternion, atan, sin / cos) stay off the serial tail: exact shortcut, like the
        // min_step clamp of select_step.  Not taken when the value itself is recorded (trace) and in Exp_SEK3's
        // theta < 1e-6 branch (translation v instead of step * v: dist ~ 1 there).
        const bool want_trace = !dry && D.trace && st->n_trace < P.trace_capacity &&
                                (k < P.trace_dense || (P.trace_every > 0 && k % P.trace_every == 0));
        const float theta_f = sqrtf(om[0] * om[0] + (om[1] * om[1] + om[2] * om[2]));
        if (!want_trace && theta_f >= 1e-6f && step * 0.9999f - 1e-6f > P.eps_2 && step <= 1.f)
          dist = (double)step;
        else
        {
          double dR[9], dT[3];
          for (int q = 0; q < 9; q++) dR[q] = dRd(q);
          for (int i = 0; i < 3; i++) dT[i] = dTd(i);
          dist = se3_log_norm(dR, dT);
        }
        const float ip_curr = (float)((double)nnz / D.sqrt_nm);  // 1486 (sqrt(N * M): IEEE, evaluated on the host)
        const bool need_decay_ell = dry ? false : indicator_update(st, sq, eq, ip_curr, P.window, P.stable_thr, e_front, s_front);
        if (dist < (double)P.eps_2) {  // CvoGPU.cu:1505-1508
          done = 1;
          st->iterations = k;
        } else {
          if (k > P.ell_decay_start && need_decay_ell) {  // CvoGPU.cu:1509-1513
            float e = ell_used * P.ell_decay_rate;
            if (e < P.ell_min) e = P.ell_min;
            st->ell = e;
            st->temp_coef = coef_of_ell(e);  // (k_coeff's per-row constant, see PairState)
          }
          st->K = min(P.K_max, (int)((double)max_nnz * 1.2));  // CvoGPU.cu:1529
          st->k = k + 1;
          if (k + 1 >= D.max_iter) {
            done = 1;
            st->iterations = k + 1;
          }
        }
      }
      st->dist = dist;
      // optional per-iteration trace (the reference's is_logging history files, CvoGPU.cu:1495-1503)
      if (!dry && D.trace && st->n_trace < P.trace_capacity &&
          (k < P.trace_dense || (P.trace_every > 0 && k % P.trace_every == 0))) {
        cvo_trace_t* tr = D.trace + st->n_trace;
        tr->k = k;
        tr->K = K_used;
        tr->ell = ell_used;
        tr->step = step;
        tr->nnz = nnz;
        tr->max_nnz = max_nnz;
        for (int q = 0; q < 3; q++) {
          tr->omega[q] = om[q];
          tr->v[q] = vv[q];
        }
        // (re-read from the staged state: eight + twelve values that would otherwise stay in registers across Exp_SEK3, the
        // pose update and the indicator just for this optional record)
        tr->B = st->B;
        tr->C = st->C;
        tr->D = st->D;
        tr->E = st->E;
        tr->dist = dist;
        for (int q = 0; q < 9; q++) tr->R[q] = st->R[q];
        for (int q = 0; q < 3; q++) tr->T[q] = st->T[q];
        st->n_trace++;
      }
    }
  }
  CVO_UPD_STAMP(3);
  // update_tf (CvoGPU.cu:94-112): the transform applied next, and the returned matrix when done
  float Ri[9], Ti[3];
  update_tf(st->R, st->T, Ri, Ti);
  {
    // Candidate-list reuse.  Target j moves by at most |Ri - Rb|_F * |y0_j| + |Ti - Tb| between the pose the
    // bitmap was built with and the one applied next.  The scan added skin_rot * rho_i + skin_tr to the cut-off
    // radius of row i, rho_i >= |y0_j| for every target that can come within the row's radius (k_prep); so as long
    // as |Ri - Rb|_F <= skin_rot and |Ti - Tb| <= skin_tr (and ell, hence every radius, has not grown) the bitmap
    // still contains every pair the exact test of k_assoc can accept.
    // (None of this reaches a result: the allowances only have to be what k_prep adds to the radii, and the motion
    // bounds carry a 0.1 % margin - hardware square roots and reciprocals, 1 ulp, instead of ~12 dependent
    // instructions per IEEE sqrtf / division on the serial tail.)
    auto fsqrt = [](float x) { return __builtin_amdgcn_sqrtf(x); };
    auto frcp = [](float x) { return __builtin_amdgcn_rcpf(x); };
    // how the last build classed the rows, the regime and the request in force: read here, once, so that the decisions
    // at the end of this block do not each start with a staging-area round trip of their own
    const int c_ovf = st->n_ovf, c_scan = st->n_scan, c_want = st->want_full;
    int c_dense = st->all_dense;
    const float ell_next = st->ell;
    const float radius = ell_next * fsqrt(fmaxf(-2.f * P.log_geo, 0.f));  // cut-off radius for l = ell
    float dr = 0, dt = 0, dr1 = 0, dt1 = 0;
    for (int q = 0; q < 9; q++) {
      const float a = Ri[q] - st->Rb[q], b = Ri[q] - st->Rinv[q];
      dr = __builtin_fmaf(a, a, dr);
      dr1 = __builtin_fmaf(b, b, dr1);
    }
    for (int q = 0; q < 3; q++) {
      const float a = Ti[q] - st->Tb[q], b = Ti[q] - st->Tinv[q];
      dt = __builtin_fmaf(a, a, dt);
      dt1 = __builtin_fmaf(b, b, dt1);
    }
    const float ymax = D.ymax;
    float rot_b = fsqrt(dr) * 1.001f, tr_b = fsqrt(dt) * 1.001f;   // since the build (the rounding slack of the two
                                                                    // transform evaluations is part of every row's skin)
    float rot_1 = fsqrt(dr1), tr_1 = fsqrt(dt1);                    // this iteration alone
    float step_move = rot_1 * ymax + tr_1;                          // what this iteration moved the farthest target
    if (P.debug_no_motion_bound) rot_b = tr_b = rot_1 = tr_1 = step_move = 0.f;  // (tests: a deliberately broken bound)
    // share of the allowances used up / used per iteration (inf when an allowance is zero and something moved)
    auto share = [&](float used, float allowance) { return used <= 0.f ? 0.f : (allowance > 0.f ? used * frcp(allowance) : __builtin_inff()); };
    const float used = fmaxf(share(rot_b, st->skin_rot), share(tr_b, st->skin_tr));
    const float rate = fmaxf(share(rot_1, st->skin_rot), share(tr_1, st->skin_tr));
    st->last_used = used;
    st->last_rate = rate;
    // the list is unusable for the coming iteration ...
    // (A list built for a larger ell stays a superset: rebuilding it after ell has shrunk only sheds candidates.  That
    // rebuild is optional, so it waits for a rebuild opportunity - flagged in the middle of a lean period it would
    // stall the pair until the next one - and, when P.shrink_align is set (CVO_SHRINK_ALIGN; off since round 6), for an
    // iteration count with (k & shrink_align) == 0: the pairs of a sub-batch decay in step, their shrink rebuilds then
    // share one pass of the rebuild kernels instead of putting real work into a different one each.)
    const bool shrink_due = ell_next < P.rebuild_shrink * st->ell_build;
    const bool shrink_now = shrink_due && trio_follows &&
                            ((st->k & P.shrink_align) == 0 || ell_next < 0.85f * P.rebuild_shrink * st->ell_build);
    bool rebuild = INIT || P.mode != 0 || !(used <= 1.f) || ell_next > st->ell_build || shrink_now;
    // ... or would expire before the next rebuild opportunity of the lean graph
    if (trio_follows && horizon > 0 && !(used + P.horizon_margin * (float)horizon * rate <= 1.f)) rebuild = true;
    // ... and, in a batch, at the common iteration counts of the optional rebuilds: a list that would not survive
    // until the next of them is renewed now, together with the other pairs' (the pass runs anyway), instead of
    // putting work into a pass of its own some opportunities later
    if (trio_follows && horizon > 0 && P.shrink_align > 0 && (st->k & P.shrink_align) == 0 &&
        !(used + P.horizon_margin * (float)(P.shrink_align + 1) * rate <= 1.f))
      rebuild = true;
    // Dense regime (rows sitting on K_max, e.g. the first iterations of an outdoor pair at a large ell): when most
    // rows overflow their lists anyway, lists are pointless - every row goes to k_assoc_dense (the reference's
    // literal ordered scan), nothing is rebuilt while that lasts, and the pair returns to lists once the rows have
    // thinned out (mean nonzeros per row below 12, far from the 32 / 64 a list holds).
    if (!INIT && P.mode == 0 && P.dense_regime) {
      const bool was = c_dense != 0;
      // (with long lists an overflow row costs what its candidates cost: the literal scan of everything only pays when
      // most rows are beyond even those, when the target cloud is small - 2048 targets are 32 lane steps, no bitmap, no
      // sort, no rebuilds: the demo pair on its K cap - or when the rows see a third of it anyway)
      const bool now = was ? (unsigned long long)st->nnz >= 12ull * (unsigned long long)D.N
                           : (2 * c_scan > D.N ||
                              (2 * c_ovf > D.N &&
                               (D.M <= 2048 || 3ull * st->ncand_list > (unsigned long long)D.N * (unsigned long long)D.M)));
      if (now != was) {
        c_dense = now ? 1 : 0;
        st->all_dense = c_dense;
        rebuild = true;
      } else if (now) {
        rebuild = false;
      }
    }
    if (rebuild) {
      for (int q = 0; q < 9; q++) st->Rb[q] = Ri[q];
      for (int q = 0; q < 3; q++) st->Tb[q] = Ti[q];
      st->ell_build = ell_next;
      // Skin: a longer-lived list costs (1 + s)^3 more candidates per iteration, a shorter-lived one more
      // rebuilds; s ~ 1.5 sqrt(step / radius) balances the two for this kernel set.  The lean graph only has a
      // rebuild opportunity every lean_U iterations, so it needs s >= ~1.3 lean_U step / radius; when that is
      // too much (fast motion) or rows overflow their lists, ask the host for the full graph.
      // 2 = a rebuild opportunity in every iteration; 4 = and k_assoc_dense (rows that overflowed the lists of the last
      // build, or the dense regime): the host has a full graph without the dense kernel for large clouds
      // (overflow rows as the LAST build left them: a pair that gains its first ones in a graph without the dense kernel
      // waits there and asks for it, see k_coeff)
      const bool dense_rows = c_ovf > 0 || c_dense != 0;
      // A wave of k_assoc runs as long as its longest row.  While a sixteenth of the rows overflow anyway (a clustered
      // cloud: k_assoc_dense runs in every iteration, its long lists cost what their candidates cost), rows of more
      // than row_max_busy candidates (8 for a few pairs in flight, 24 up to 16 pairs, none beyond: a full chip keeps its rows here) join them - a wave per row, 64 candidates
      // per step - instead of holding 63 neighbours back.  (Free to follow the launch: no result depends on a row's class.)
      st->row_max = (!INIT && P.long_lists && !c_dense && 16 * c_ovf > D.N) ? P.row_max_busy : ASSOC_CAP16;
      if (P.long_lists && P.row_max_cap > 0) st->row_max = min(st->row_max, P.row_max_cap);
      int want_full = 2;
      float s = 0.f;
      // rows beyond every list fall back to the literal scan over all targets (k_assoc_dense): fine for a few
      // rows or a small cloud, ruinous if a generous skin pushes many rows of a large one over the edge - the skin
      // backs off by halves while the last build left such rows and recovers slowly afterwards
      if (INIT) st->skin_scale = 1.f;
      else if (c_scan > 0 && !c_dense) st->skin_scale = fmaxf(0.5f * st->skin_scale, 1.f / 64.f);
      else st->skin_scale = fminf(1.f, 1.1f * st->skin_scale);
      if (!INIT && P.mode == 0 && P.use_geo && radius > 0.f && P.skin_frac > 0.f && !c_dense) {
        const float rel = step_move * frcp(radius);
        s = st->skin_scale * P.skin_frac * fminf(fmaxf(1.5f * fsqrt(rel), P.skin_min), P.skin_max);
        const float s_lean = fmaxf(s, P.lean_skin * (float)P.lean_U * rel);
        const float s_lean2 = fmaxf(s, P.lean_skin * (float)P.lean_U2 * rel);
        // (rows that walk long lists cost what their candidates cost, whatever the skin; rows scanned literally do not)
        if (s_lean <= 0.5f && c_scan == 0) {
          s = s_lean;
          want_full = 0;
        } else if (P.lean_U2 > 0 && s_lean2 <= 0.5f && c_scan == 0) {
          s = s_lean2;  // too fast for lean_U iterations between rebuilds, slow enough for lean_U2
          want_full = 1;
        } else if (!(s >= 2.f * rel)) {
          s = 0.f;  // would not survive two iterations: plain scan every iteration
        }
      }
      if (!(s == s)) s = 0.f;
      // s * radius is what the FARTHEST target may move; split into a rotation and a translation allowance in the
      // proportion of the current motion (plus a blend of the pooled budget for either, so that a change of
      // direction does not expire the lists at once): every row's skin follows from its own distance (k_prep)
      {
        // (normalised so that the farthest row gets exactly s * radius)
        const float life = step_move > 0.f ? s * radius * frcp(step_move * (1.f + P.skin_blend)) : 0.f;  // iterations at the current speed
        const float bl = P.skin_blend;
        st->skin_rot = life * ((1.f - bl) * rot_1 + bl * step_move * frcp(fmaxf(ymax, 1e-20f)));
        st->skin_tr = life * ((1.f - bl) * tr_1 + bl * step_move);
        if (!(st->skin_rot == st->skin_rot) || !(st->skin_tr == st->skin_tr)) st->skin_rot = st->skin_tr = 0.f;
      }
      if (c_dense) want_full = -1;  // dense regime: nothing is rebuilt until the pair leaves it
      want_full = want_encode(want_full, dense_rows);
      st->want_full = want_full;
      if (!dry) out.post_want(D, spec, want_full);
      st->n_builds = INIT ? 1 : st->n_builds + 1;
      st->rebuild = 1;  // cleared by k_list once bitmap and lists are current
    } else if (c_scan == 0 && !c_dense) {  // has the motion slowed down enough for a leaner graph?
      const float c = fminf(P.lean_skin, 1.3f);
      int want = want_level(c_want);
      if (used + c * (float)P.lean_U * rate <= 1.f)
        want = 0;
      else if (P.lean_U2 > 0 && used + c * (float)P.lean_U2 * rate <= 1.f)
        want = min(want, 1);
      // -1 = calm: at the current speed the list outlives P.calm_U more iterations - the host may run this pair on the
      // lean graph with ONE rebuild opportunity per chunk (the opportunities are three launches each, and in the end
      // game - the step clamped at min_step, rebuilds only when ell has decayed - nearly all of them find nothing to do)
      if (want == 0 && P.calm_U > 0 && used + c * (float)P.calm_U * rate <= 1.f) want = -1;
      want = want_encode(want, c_ovf > 0);
      if (want != c_want) {
        st->want_full = want;
        if (!dry) out.post_want(D, spec, want);
      }
    }
  }
  CVO_UPD_STAMP(4);
  for (int q = 0; q < 9; q++) st->Rinv[q] = Ri[q];
  for (int q = 0; q < 3; q++) st->Tinv[q] = Ti[q];
  if (done || INIT || P.mode != 0) {  // the returned matrix (final update_tf, CvoGPU.cu:1562): only read once the pair is done
    for (int i = 0; i < 3; i++) {
      for (int j = 0; j < 3; j++) st->out_T[4 * j + i] = Ri[3 * i + j];
      st->out_T[12 + i] = Ti[i];
    }
    st->out_T[3] = st->out_T[7] = st->out_T[11] = 0;
    st->out_T[15] = 1;
  }
  if (clk0 && !dry) {  // CVO_KERNEL_CLOCK (k_coeff): this launch's interval and the association's, see PairState
    if (st->clk_last_assoc) {
      st->clk_sum[0] += st->clk_last_assoc;
      st->clk_n[0]++;
      st->clk_last_assoc = 0;
    }
    const unsigned dt_coeff = pair_clock_ticks(clk0);
    if (dt_coeff) {
      st->clk_sum[1] += dt_coeff;
      st->clk_n[1]++;
    }
  }
  if (done && !dry) {
    st->status = 1;
    out.post_done(D, spec);
  }
}

// The steps a speculative run is worth: the clamp values of compute_step_size (CvoGPU.cu:1153-1158) - the end game of every
// BASELINE shape sits on min_step for ~95 % of its iterations, the demo pair on max_step for its first hundred.
__device__ __forceinline__ bool step_is_clamp(const DevParams& P, float step) {
  return step > 0.f && (step == P.min_step || step == P.max_step);
}

// nnz / max_nnz / candidates / overflow rows of the iteration from k_assoc's block counts (an earlier kernel: plain loads),
// reduced in update_body's order: lane l of a row of 16 takes blocks l, l + 16, ...; component c lives in row c.
// Lane 0 leaves the four results in s_n.  Called by a whole wave.
__device__ __forceinline__ void reduce_counts(const unsigned long long* cnt_part, int nba, int tid, unsigned long long* s_n) {
  const CVO_GLOBAL unsigned* cnt32 = as_global(reinterpret_cast<const unsigned*>(cnt_part));
  const int c = tid >> 4, bl = tid & 15;
  auto addsat = [](unsigned a, unsigned b) { const unsigned r = a + b; return r < a ? 0xffffffffu : r; };
  unsigned q = 0;
  for (int b0 = bl; b0 < nba; b0 += 128) {
    unsigned v[8];
#pragma unroll
    for (int u = 0; u < 8; u++) {
      const int b = b0 + 16 * u;
      v[u] = b < nba ? ld_g<false>(cnt32 + ((size_t)b * 4 + c) * 2) : 0u;
    }
#pragma unroll
    for (int u = 0; u < 8; u++) q = (c == 1) ? max(q, v[u]) : addsat(q, v[u]);
  }
  auto meet = [&](unsigned o) { q = (c == 1) ? max(q, o) : addsat(q, o); };
  meet((unsigned)dpp_i32<DPP_XOR1>((int)q));
  meet((unsigned)dpp_i32<DPP_XOR2>((int)q));
  meet((unsigned)dpp_i32<DPP_HALF_MIRROR>((int)q));
  meet((unsigned)dpp_i32<DPP_MIRROR>((int)q));
#pragma unroll
  for (int cc = 0; cc < 4; cc++) {
    const unsigned long long qv = (unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)q, 16 * cc);
    if (tid == 0) s_n[cc] = qv;
  }
}

// The speculative run (round 6).  One extra block per pair and k_coeff launch - its first wave - runs update_advance on the
// state as the launch found it with the PREVIOUS step, while the row blocks walk their nonzeros: everything update_advance
// needs but the step is known when k_coeff starts (the twist and the counts are k_assoc's, the indicator depends on the
// nonzero count only).  The result - the staged state and UpdOut - is published as data-tagged granules (cvo_wave.h);
// update_body adopts it when the step it derives from B..E is bit for bit the predicted one (~95 % of the iterations of a
// BASELINE shape: the clamp at min_step), and computes as before otherwise.  Same function, same inputs, same order:
// the state a pair ends an iteration with does not depend on who computed it.  (The indicator FIFO entries the run pushes
// are the ones update_advance would push whatever the step: written twice with the same value when the run is not adopted.)
__device__ __forceinline__ void update_speculate(const PairDesc* __restrict__ Dp, PairState* const gst, const DevParams& P, int flags,
                                                 UpdateShared& U, unsigned long long call_serial) {
  const int tid = threadIdx.x;  // < 64
  // (gst == D.st from the kernel's state array: the state is requested next to the descriptor, not behind it - this run
  // has the row blocks' time and no more)
  const UpdDesc D = load_upd_desc(Dp);
  unsigned* const s_hot = U.hot;
  {
    const unsigned h0 = reinterpret_cast<const unsigned*>(gst)[tid];
    const unsigned h1 = (tid + 64 < HOT_DWORDS) ? reinterpret_cast<const unsigned*>(gst)[tid + 64] : 0u;
    s_hot[tid] = h0;
    if (tid + 64 < HOT_DWORDS) s_hot[tid + 64] = h1;
  }
  __builtin_amdgcn_wave_barrier();
  PairState* const st = reinterpret_cast<PairState*>(s_hot);
  const float pred = st->step;
  if (!step_is_clamp(P, pred)) return;
  // Is this the state the launch STARTED with?  The block is the last of its pair to be dispatched; on a full chip it can
  // start after the pair's row blocks have finished and the update has run - the state it stages is then the NEXT
  // iteration's, the twist and counts still this one's, and the tag it would publish under the next launch's.  The flow gate
  // stamped the twist with the (launch generation, iteration) k_assoc saw: anything else in the state means "late" (or a
  // state caught in the middle of the update's write-back) and the run is abandoned before it touches anything.
  {
    const int xe = __float_as_int(gst->xi[46]), xk = __float_as_int(gst->xi[47]);
    if (xe != st->epoch || xk != st->k || st->status != 0) return;
  }
  // (the tag of this launch's partials, from the VALIDATED generation: k_coeff's row blocks form the same one)
  const unsigned tag = partial_tag(call_serial, 0x80000000u | (unsigned)st->epoch);
  reduce_counts(D.cnt_part, Dp->nblk_assoc, tid, U.n);
  __builtin_amdgcn_wave_barrier();
  if (tid == 0) {
    const float e_front = gst->eq[st->e_head], s_front = gst->sq[st->s_head];
    const XiMats* xm = reinterpret_cast<const XiMats*>(gst->xi);  // the twist of this iteration (k_assoc's flow gate)
    const float twist[6] = {xm->omega[0], xm->omega[1], xm->omega[2], xm->v[0], xm->v[1], xm->v[2]};
    UpdOut out;
    update_advance<false>(D, P, flags, st, gst->sq, gst->eq, nullptr, U.n, pred, twist, e_front, s_front, 0ull, true, out);
    U.ext[0] = (unsigned)out.done;
    U.ext[1] = (unsigned)out.want_write;
    U.ext[2] = (unsigned)out.want_val;
    U.ext[3] = __float_as_uint(pred);
    for (int q = 4; q < SHADOW_WORDS - HOT_DWORDS; q++) U.ext[q] = 0u;
  }
  __builtin_amdgcn_wave_barrier();
  const unsigned long long t = (unsigned long long)tag << 32;
  const unsigned w0 = s_hot[tid], w1 = (tid + 64 < HOT_DWORDS) ? s_hot[tid + 64] : U.ext[tid + 64 - HOT_DWORDS];
  __hip_atomic_store(D.shadow + tid, t | w0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_store(D.shadow + 64 + tid, t | w1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <bool INIT, bool COH>
__device__ __forceinline__ void update_body(const UpdDesc& D, const DevParams& P, int flags,
                                            int n_flow_parts, UpdateShared& U, const float* twist,
                                            const unsigned* preloaded_hot, unsigned long long clk0 = 0ull, unsigned tag = 0u,
                                            bool may_adopt = false) {
  PairState* const gst = D.st;
  const bool dry = (flags & 8) != 0;  // timing replay: compute everything, write nothing back
  double* const s_c = U.c;
  unsigned long long* const s_n = U.n;
  unsigned* const s_hot = U.hot;
  const int tid = threadIdx.x;
  const bool act = tid < 64;
  CVO_UPD_STAMP(0);

  // the scalar part of the state is staged through LDS: one coalesced burst in, one out, instead of
  // dozens of dependent global accesses from a single lane
  static_assert(HOT_DWORDS <= 128, "two dwords per lane of the first wave cover the scalar state");
  if (act && preloaded_hot) {  // the caller read the state into registers while it waited for something else
    s_hot[tid] = preloaded_hot[0];
    if (tid + 64 < HOT_DWORDS) s_hot[tid + 64] = preloaded_hot[1];
  } else if (act) {
    for (int q = tid; q < HOT_DWORDS; q += 64) s_hot[q] = reinterpret_cast<const unsigned*>(gst)[q];
  }
  PairState* const st = reinterpret_cast<PairState*>(s_hot);
  float* const sq = gst->sq;
  float* const eq = gst->eq;
  // Speculation (update_speculate): when the previous step sat on a clamp, another block of this launch has run everything
  // that follows the step with that same step while the row blocks worked; if the step of this iteration turns out to be the
  // predicted one, its state is adopted instead of computed here - on this pair's serial chain.  The shadow's granules are
  // requested now, next to the partials.
  bool spec_try = false;
  float pred = 0.f;
  unsigned long long g0 = 0ull, g1 = 0ull;
  if (!INIT && COH && may_adopt && !dry && P.mode == 0 && P.trace_capacity == 0) {
    __builtin_amdgcn_wave_barrier();  // (the staged state: LDS writes of this wave, read back below)
    pred = st->step;
    spec_try = step_is_clamp(P, pred);
    if (spec_try) {
      g0 = ld_g<true>(as_global(D.shadow) + tid);
      g1 = ld_g<true>(as_global(D.shadow) + 64 + tid);
    }
  }
  if (!INIT && act) {
    // The four thrust::reduce of compute_step_size (CvoGPU.cu:1118-1121) and the nonzero / max counts
    // (SparseKernelMat.cu:37-46, CvoGPU.cu:1518): lane l owns component (l & 3) of blocks l>>2, l>>2 + 16, ...
    // so all loads are in flight at once; a fixed xor-shuffle tree finishes (deterministic order).
    // Row c of 16 lanes owns component c; lane l of the row takes blocks l, l + 16, ... (eight loads in flight), the
    // row meets through a DPP butterfly (no LDS) and every lane reads the four results with v_readlane.
    const int nba = n_flow_parts, nbc = D.nblk_coeff;
    const int c = tid >> 4, bl = tid & 15;
    double s = 0;
    // the counts were written by the association kernel(s), i.e. before this launch: plain loads, requested ahead of
    // the coherent ones so that the two round trips overlap
    // (every count of one iteration fits 32 bits - at most rows x K_max nonzeros, rows x targets candidates, checked
    // at set-up - and the block partials are read as such: the 64-bit DPP steps cost four times the instructions)
    // (global, not flat, addresses: the two groups of loads below are then really in flight together - a flat load makes
    // the compiler wait for vmcnt AND lgkmcnt to drain before anything that follows it)
    const CVO_GLOBAL unsigned* cnt32 = as_global(reinterpret_cast<const unsigned*>(D.cnt_part));
    const CVO_GLOBAL unsigned long long* coef_part = as_global(D.coef_part);
    unsigned vq[8];
#pragma unroll
    for (int u = 0; u < 8; u++) {
      const int b = bl + 16 * u;
      vq[u] = b < nba ? ld_g<false>(cnt32 + ((size_t)b * 4 + c) * 2) : 0u;
    }
    if (P.mode == 0) {
      // four blocks (eight coherent granule loads) in flight per lane and round (64 row-block slices = one round), summed in
      // block order; a granule that does not carry this launch's tag yet has not landed: the round is read again (cvo_wave.h)
      for (int b0 = bl; b0 < nbc; b0 += 64) {
        TaggedF64 v[4];
        for (int polls = 0;; polls++) {
          bool ok = true;
#pragma unroll
          for (int u = 0; u < 4; u++) {
            const int b = min(b0 + 16 * u, nbc - 1);
            v[u] = ld_tagged<COH>(coef_part + (size_t)b * COEF_GRANULES + 2 * c);
          }
#pragma unroll
          for (int u = 0; u < 4; u++) ok = ok && (!COH || v[u].carries(tag));
          if (__ballot(!ok) == 0ull) break;
          if (polls > PARTIAL_POLL_LIMIT) {  // (bounded, see coeff_twist_load)
            if (tid == 0) st->sync_err = 2;  // (the staged state: update_advance ends the pair)
            break;
          }
          __builtin_amdgcn_s_sleep(2);
        }
#pragma unroll
        for (int u = 0; u < 4; u++) s += (b0 + 16 * u < nbc) ? v[u].value() : 0.0;
      }
    } else if (c == 0) {
      for (int b = bl; b < nba; b += 16) s += ld_tagged<false>(as_global(D.flow_part) + (size_t)b * FLOW_GRANULES + 12).value();
    }
    // (component 2, the candidate statistic, can exceed 32 bits for very large clouds before the dense regime engages:
    // it saturates instead of wrapping; nnz / overflow rows are bounded by the set-up check)
    auto addsat = [](unsigned a, unsigned b) { const unsigned r = a + b; return r < a ? 0xffffffffu : r; };
    unsigned q = 0;
#pragma unroll
    for (int u = 0; u < 8; u++) q = (c == 1) ? max(q, vq[u]) : addsat(q, vq[u]);
    for (int b0 = bl + 128; b0 < nba; b0 += 128) {
      unsigned v[8];
#pragma unroll
      for (int u = 0; u < 8; u++) {
        const int b = b0 + 16 * u;
        v[u] = b < nba ? ld_g<false>(cnt32 + ((size_t)b * 4 + c) * 2) : 0u;
      }
#pragma unroll
      for (int u = 0; u < 8; u++) q = (c == 1) ? max(q, v[u]) : addsat(q, v[u]);
    }
    s += dpp_f64<DPP_XOR1>(s);
    s += dpp_f64<DPP_XOR2>(s);
    s += dpp_f64<DPP_HALF_MIRROR>(s);
    s += dpp_f64<DPP_MIRROR>(s);
    {
      auto meet = [&](unsigned o) { q = (c == 1) ? max(q, o) : addsat(q, o); };
      meet((unsigned)dpp_i32<DPP_XOR1>((int)q));
      meet((unsigned)dpp_i32<DPP_XOR2>((int)q));
      meet((unsigned)dpp_i32<DPP_HALF_MIRROR>((int)q));
      meet((unsigned)dpp_i32<DPP_MIRROR>((int)q));
    }
#pragma unroll
    for (int cc = 0; cc < 4; cc++) {
      const double sv = lane_f64(s, 16 * cc);
      const unsigned long long qv = (unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)q, 16 * cc);
      if (tid == 0) {
        s_c[cc] = sv;
        s_n[cc] = qv;
      }
    }
  }
  __syncthreads();
  // fronts of the indicator FIFOs (HBM), on their way while the step is computed
  float e_front = 0.f, s_front = 0.f;
  if (!INIT && tid == 0) {
    e_front = eq[st->e_head];
    s_front = sq[st->s_head];
  }
  CVO_UPD_STAMP(1);
  // the step of this iteration: the cubic's real roots are searched on three lanes side by side
  float step_w = 0.f;
  if (!INIT && act && P.mode == 0) step_w = select_step<true>(s_c[0], s_c[1], s_c[2], s_c[3], P.min_step, P.max_step);
  CVO_UPD_STAMP(2);
  UpdOut out;
  bool adopt = false;
  if (spec_try && __float_as_uint(step_w) == __float_as_uint(pred) && s_hot[offsetof(PairState, sync_err) / 4] == 0u) {
    // every granule of the shadow carries this launch's tag <=> the speculative run has finished and published
    for (int polls = 0;; polls++) {
      const bool ok = (unsigned)(g0 >> 32) == tag && (unsigned)(g1 >> 32) == tag;
      if (__ballot(!ok) == 0ull) {
        adopt = true;
        break;
      }
      if (polls >= 3) break;  // (late: computing here is cheaper than waiting)
      __builtin_amdgcn_s_sleep(8);
      g0 = ld_g<true>(as_global(D.shadow) + tid);
      g1 = ld_g<true>(as_global(D.shadow) + 64 + tid);
    }
  }
  if (adopt) {
    s_hot[tid] = (unsigned)g0;
    if (tid + 64 < HOT_DWORDS)
      s_hot[tid + 64] = (unsigned)g1;
    else
      U.ext[tid + 64 - HOT_DWORDS] = (unsigned)g1;
    __builtin_amdgcn_wave_barrier();
    if (tid == 0) {
      st->B = s_c[0];
      st->C = s_c[1];
      st->D = s_c[2];
      st->E = s_c[3];
      st->n_adopted++;
      if (U.ext[1]) {
        *D.want_out = (int)U.ext[2];
        *D.want_host = (int)U.ext[2];
      }
      if (U.ext[0]) {
        *D.status_out = 1;
        *D.status_host = 1;
      }
    }
  } else if (tid == 0) {
    update_advance<INIT>(D, P, flags, st, sq, eq, s_c, s_n, step_w, twist, e_front, s_front, clk0, false, out);
  }
  CVO_UPD_STAMP(5);
  __syncthreads();
  if (act && !dry)
    for (int q = tid; q < HOT_DWORDS; q += 64) reinterpret_cast<unsigned*>(gst)[q] = s_hot[q];
  CVO_UPD_STAMP(6);
}

template <bool INIT>
__global__ __launch_bounds__(64) void k_update(const PairDesc* __restrict__ descs, const DevParams* __restrict__ Pp,
                                               const int* __restrict__ status, int flags) {
  if (!INIT && status[blockIdx.x] != 0) return;
  const PairDesc* __restrict__ D = descs + blockIdx.x;
  if (!INIT && (flags & 1) && (D->st->rebuild || (D->st->n_ovf > 0 && !(flags & 32)))) return;  // lean graph: the pair is waiting (k_assoc)
  if (INIT && threadIdx.x == 0) {  // the pair's cross-block counters start at zero
    *D->status_out = 0;   // (a slot of a batch queue: the words of its previous occupant say "finished")
    *D->status_host = 0;
    *D->gate = 0;
    *D->gate_flow = 0;
    *D->done = 0;
    *D->tile_count = 0ull;
  }
  const DevParams P = *Pp;
  __shared__ UpdateShared U;
  update_body<INIT, false>(load_upd_desc(D), P, flags, D->nblk_assoc, U, nullptr, nullptr);
}

}  // namespace cvo_dev
