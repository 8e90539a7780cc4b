// cvo::ShardPlan: the index arithmetic of the batched multi-GPU mode, free of any device code so
// that it can be tested without GPUs.  n independent frame pairs over D devices in contiguous blocks of
// per = ceil(n / D) pairs: pair p runs on device p / per (64 per GPU for BASELINE.json configs[4]: 512 pairs, 8 GPUs);
// every device contributes exactly `per` slots to the all-gather (the last devices' unused slots hold identity poses),
// so the gathered array is rank-major with `per` slots per rank and pair p is found at slot device_of(p) * per +
// (p - lo(device_of(p))) = p.
#pragma once
#include <algorithm>

namespace cvo {

struct ShardPlan {
  int n = 0, D = 1, per = 0;
  ShardPlan(int n_pairs, int n_devices) : n(std::max(n_pairs, 0)), D(std::max(n_devices, 1)) { per = (n + D - 1) / D; }
  int lo(int d) const { return std::min(n, d * per); }            // first pair of device d
  int hi(int d) const { return std::min(n, lo(d) + per); }        // one past its last pair (== lo: the device idles)
  int count(int d) const { return hi(d) - lo(d); }
  int device_of(int p) const { return per > 0 ? p / per : 0; }
  int local_index(int p) const { return p - lo(device_of(p)); }   // index of pair p inside its device's batch
  int slot_of(int p) const { return device_of(p) * per + local_index(p); }  // index in the gathered [D x per] array
  int gathered_slots() const { return D * per; }
};

}  // namespace cvo
